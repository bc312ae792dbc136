function [vp0_vec,vp0_type,elcbo_beta,compute_var,NSentK,NSentKFast] = vpsieve_vbmc(Ninit,Nbest,vp,gp,optimState,options,K)
%VPSIEVE_VBMC Drop-in shim: the preliminary 'sieve' of variational-posterior candidates with ALL candidates
% evaluated in one batched pass on an MI355X (vbmc_hip_mex 'elbo_batch').
%
% Same signature and defaulting as the reference (misc/vpsieve_vbmc.m:1).  Written from the batched host mirror
% vbmc_amd/optimize.py:vpsieve_vbmc / sieve_evaluate: counts and the confidence weight from the options, candidates
% from the user's own vbinit_vbmc / gethpd_vbmc / vpbounds (host bookkeeping, not shadowed), then
%     nelcbo_fill = vbmc_hip_sieve(...)      one device pass instead of numel(vp0_vec) calls of negelcbo_vbmc
% and the stable ascending sort.  Everything outside the accelerated path (unsupported mean function, integrated mean,
% output warping, vp.delta ~= 0, ...) goes to the reference implementation further down the path, decided BEFORE any
% random number is drawn so that the fall-through consumes the same stream the reference would.
if nargin < 7 || isempty(K); K = vp.K; end
if nargin < 5; optimState = []; end
if ~isfield(optimState,'delta'); optimState.delta = 0; end
vpchk = vp; vpchk.K = K; vpchk.delta = optimState.delta;
if ~vbmc_hip_supported(gp,vpchk)
    ref = vbmc_hip_reference('vpsieve_vbmc');
    [vp0_vec,vp0_type,elcbo_beta,compute_var,NSentK,NSentKFast] = ref(Ninit,Nbest,vp,gp,optimState,options,K);
    return;
end
if ~isfield(optimState,'EntropySwitch'); optimState.EntropySwitch = false; end
if ~isfield(optimState,'Neff'); optimState.Neff = size(gp.X,1); end
if isempty(Nbest); Nbest = 1; end

vp.delta = optimState.delta(:);
if isempty(Ninit); Ninit = ceil(evaloption_vbmc(options.NSelbo,K)); end

% samples per component for the MC entropy: optimisation / preliminary evaluation
NSentK = ceil(evaloption_vbmc(options.NSent,K)/K);
NSentKFast = ceil(evaloption_vbmc(options.NSentFast,K)/K);
if optimState.EntropySwitch || K == 1
    NSentK = 0; NSentKFast = 0;
end
elcbo_beta = evaloption_vbmc(options.ELCBOWeight,optimState.Neff);
compute_var = elcbo_beta ~= 0;

[vp,thetabnd] = vpbounds(vp,gp,options,K);

if Ninit > 0
    [Xstar,ystar] = gethpd_vbmc(gp.X,gp.y,options.HPDFrac);
    if Nbest == 1
        [vp0_vec,vp0_type] = vbinit_vbmc(1,Ninit,vp,K,Xstar,ystar);
    else
        n3 = ceil(Ninit/3);
        [va,ta] = vbinit_vbmc(1,n3,vp,K,Xstar,ystar);
        [vb,tb] = vbinit_vbmc(2,n3,vp,K,Xstar,ystar);
        [vc,tc] = vbinit_vbmc(3,Ninit-2*n3,vp,K,Xstar,ystar);
        vp0_vec = [va,vb,vc];
        vp0_type = [ta;tb;tc];
    end
    if isfield(optimState,'vp_repo') && ~isempty(optimState.vp_repo) && options.VariationalInitRepo
        Ntheta = numel(get_vptheta(vp0_vec(1)));
        for ii = 1:numel(optimState.vp_repo)
            if numel(optimState.vp_repo{ii}) == Ntheta
                vp0_vec = [vp0_vec,rescale_params(vp0_vec(1),optimState.vp_repo{ii})]; %#ok<AGROW>
                vp0_type = [vp0_type;1]; %#ok<AGROW>
            end
        end
    end
    % rescaled candidates (what the loop of the reference leaves in vp0_vec), then one batched evaluation
    for iOpt = 1:numel(vp0_vec)
        [~,vp0_vec(iOpt)] = get_vptheta(vp0_vec(iOpt),vp.optimize_mu,vp.optimize_sigma,vp.optimize_lambda,vp.optimize_weights);
    end
    nelcbo_fill = vbmc_hip_sieve(vp0_vec,gp,NSentKFast,compute_var,elcbo_beta,thetabnd);
    [~,vp0_ord] = sort(nelcbo_fill(:),'ascend');       % stable, NaN last: index-identical to the sequential loop
    vp0_vec = vp0_vec(vp0_ord);
    vp0_type = vp0_type(vp0_ord);
else
    vp0_vec = vp;
    vp0_type = 1;
end
end
