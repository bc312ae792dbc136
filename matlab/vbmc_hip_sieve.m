function score = vbmc_hip_sieve(calls,gp,elcbo_beta)
%VBMC_HIP_SIEVE The recorded sieve candidates in ONE batched device pass.
%   SCORE = VBMC_HIP_SIEVE(CALLS,GP,ELCBO_BETA): CALLS is the cell array the negelcbo_vbmc shim recorded while the
%   reference's misc/vpsieve_vbmc.m:74-78 ran (one struct per candidate: theta, vp, Ns, compute_var, thetabnd);
%   SCORE(i) = nelbo_i + ELCBO_BETA*sqrt(varF_i), what the reference's loop would have stored in nelcbo_fill(i).
% Candidates share K and the optimize_* flags (vbinit_vbmc builds them from one vp); those whose NON-optimised groups differ
% go in separate sub-batches (those groups are not part of theta).  A candidate with a non-finite parameter is not sent
% (the library validates the whole batch): its score is NaN, as the reference's arithmetic would give, and sorts last.
% In a multi-device session the candidates of a sub-batch are dealt over the devices (r = g mod n) and their values
% all-gathered over xGMI inside the library ('elbo_batch_multi'): the estimator of the one-device pass sample for sample, equal to
% it to the order of summation (1e-13), identical on every device of the session.
% If the device refuses a sub-batch (vbmc_hip:unsupported) its members are evaluated one by one through the
% negelcbo_vbmc shim, which falls through to the reference on its own.
R = numel(calls);
score = NaN(1,R);
c1 = calls{1};
T = numel(c1.theta);
Theta = zeros(T,R);
label = cell(1,R);
for i = 1:R
    Theta(:,i) = calls{i}.theta;
    label{i} = fixed_label(calls{i}.vp);
end
finite = all(isfinite(Theta),1);
[~,~,grp] = unique(label,'stable');
ndev = vbmc_hip_mex('comm_size');                                      % > 1: VBMC_HIP_DEVICES / 'comm_open' (every GPU of the node)
if ndev > 1; h = vbmc_hip_gp_handle(gp,'all'); batch = 'elbo_batch_multi'; else; h = vbmc_hip_gp_handle(gp); batch = 'elbo_batch'; end
for g = 1:max(grp)
    members = find(grp(:)' == g & finite);
    if isempty(members); continue; end
    seed = 0;
    if c1.Ns > 0; seed = randi(2^31-1); end                           % device stream of this sub-batch
    try
        [F,~,varG] = vbmc_hip_mex(batch,h,Theta(:,members),calls{members(1)}.vp,c1.Ns,0,double(c1.compute_var), ...
            0,c1.thetabnd,seed);
        if ~c1.compute_var; varG = zeros(size(F)); end
        score(members) = F + elcbo_beta*sqrt(varG);
    catch err
        if ~strcmp(err.identifier,'vbmc_hip:unsupported'); rethrow(err); end
        for i = members
            [f,~,~,~,v] = negelcbo_vbmc(Theta(:,i),0,calls{i}.vp,gp,c1.Ns,0,c1.compute_var,0,c1.thetabnd);
            score(i) = f + elcbo_beta*sqrt(v);
        end
    end
end
end

function s = fixed_label(v)
fx = [];
if ~v.optimize_mu; fx = [fx; v.mu(:)]; end
if ~v.optimize_sigma; fx = [fx; v.sigma(:)]; end
if ~v.optimize_lambda; fx = [fx; v.lambda(:)]; end
if ~v.optimize_weights; fx = [fx; v.w(:)]; end
s = sprintf('%.17g,',fx);
end
