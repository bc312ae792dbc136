function nelcbo_fill = vbmc_hip_sieve(vp0_vec,gp,NSentKFast,compute_var,elcbo_beta,thetabnd)
%VBMC_HIP_SIEVE All sieve candidates in ONE batched device pass.
%
% Replaces the sequential loop of misc/vpsieve_vbmc.m:74-78
%     for iOpt = 1:Nopts
%         [theta0,vp0_vec(iOpt)] = get_vptheta(vp0_vec(iOpt), ...);
%         [nelbo_tmp,~,~,~,varF_tmp] = negelcbo_vbmc(theta0,0,vp0_vec(iOpt),gp,NSentKFast,0,compute_var,...,thetabnd);
%         nelcbo_fill(iOpt) = nelbo_tmp + elcbo_beta*sqrt(varF_tmp);
%     end
% by   nelcbo_fill = vbmc_hip_sieve(vp0_vec,gp,NSentKFast,compute_var,elcbo_beta,thetabnd);
% The caller keeps its own [~,vp0_ord] = sort(nelcbo_fill,'ascend') (:82), so the order is index-identical.
% Candidates must share K and the optimize_* flags (they do: vbinit_vbmc builds them from one vp);
% non-optimised groups that differ between candidates are evaluated in sub-batches.
R = numel(vp0_vec);
nelcbo_fill = zeros(1,R);
T = numel(get_vptheta(vp0_vec(1)));
Theta = zeros(T,R);
key = cell(1,R);
for i = 1:R
    [Theta(:,i),vp0_vec(i)] = get_vptheta(vp0_vec(i));
    v = vp0_vec(i); fx = [];
    if ~v.optimize_mu; fx = [fx; v.mu(:)]; end %#ok<AGROW>
    if ~v.optimize_sigma; fx = [fx; v.sigma(:)]; end %#ok<AGROW>
    if ~v.optimize_lambda; fx = [fx; v.lambda(:)]; end %#ok<AGROW>
    if ~v.optimize_weights; fx = [fx; v.w(:)]; end %#ok<AGROW>
    key{i} = sprintf('%.17g,',fx);
end
h = vbmc_hip_gp_handle(gp);
[~,~,grp] = unique(key,'stable');
for g = 1:max(grp)
    idx = find(grp == g);
    [F,~,varG] = vbmc_hip_mex('elbo_batch',h,Theta(:,idx),vp0_vec(idx(1)),NSentKFast,0,double(compute_var), ...
        0,thetabnd,randi(2^31-1));
    if compute_var; nelcbo_fill(idx) = F + elcbo_beta*sqrt(varG); else; nelcbo_fill(idx) = F; end
end
end
