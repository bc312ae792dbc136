function [vp,varss,pruned] = vpoptimize_vbmc(Nfastopts,Nslowopts,vp,gp,K,optimState,options,prnt)
%VPOPTIMIZE_VBMC Drop-in shim: optimisation of the variational posterior with the Nslowopts Adam chains run in
% lock-step ENTIRELY on an MI355X (vbmc_hip_mex 'adam') and their 2*Nslowopts full-ELCBO evaluations in one batched pass.
%
% Same signature and defaulting as the reference (misc/vpoptimize_vbmc.m:1).  Written from the batched host mirror
% vbmc_amd/optimize.py:vpoptimize_vbmc: (1) batched sieve (matlab/vpsieve_vbmc.m), (2) the starting points picked by
% candidate type, (3) all chains through ONE on-device Adam loop (utils/fminadam.m per chain, including the
% 20-iteration stopping test), (4) one batched evaluation of the full ELCBO (fine MC entropy, full variance,
% per-component terms) at every chain's best midpoint and endpoint, (5) the best slot by ELCBO, (6) component pruning,
% one evaluation at a time because each depends on the last.
%
% Falls through to the reference further down the path, BEFORE any random number is drawn, for everything outside that
% path: deterministic entropy (NSentK = 0: fminunc), ELCBOWeight ~= 0 or StochasticOptimizer ~= 'adam' (CMA-ES),
% unsupported surrogates (vbmc_hip_supported).
if nargin < 5 || isempty(K); K = vp.K; end
if nargin < 6; optimState = []; end
if nargin < 7; options = []; end
if nargin < 8 || isempty(prnt); prnt = 0; end

fallthrough = isempty(options);
if ~fallthrough
    if ~isfield(optimState,'delta'); optimState.delta = 0; end
    if ~isfield(optimState,'EntropySwitch'); optimState.EntropySwitch = false; end
    if ~isfield(optimState,'Neff'); optimState.Neff = size(gp.X,1); end
    vpchk = vp; vpchk.K = K; vpchk.delta = optimState.delta;
    NSentK = ceil(evaloption_vbmc(options.NSent,K)/K);
    if optimState.EntropySwitch || K == 1; NSentK = 0; end
    elcbo_beta = evaloption_vbmc(options.ELCBOWeight,optimState.Neff);
    fallthrough = NSentK == 0 || elcbo_beta ~= 0 || ~strcmpi(options.StochasticOptimizer,'adam') ...
        || ~vbmc_hip_supported(gp,vpchk);
end
if fallthrough
    ref = vbmc_hip_reference('vpoptimize_vbmc');
    [vp,varss,pruned] = ref(Nfastopts,Nslowopts,vp,gp,K,optimState,options,prnt);
    return;
end
if ~isfield(optimState,'Warmup'); optimState.Warmup = ~vp.optimize_weights; end
if ~isfield(optimState,'temperature'); optimState.temperature = 1; end

% (1) batched sieve
[vp0_vec,vp0_type,elcbo_beta,compute_var,NSentK] = vpsieve_vbmc(Nfastopts,Nslowopts,vp,gp,optimState,options,K);
[vp,thetabnd] = vpbounds(vp,gp,options,K);

% (2) starting points by candidate type
for iOpt = 1:Nslowopts
    if Nslowopts == 1
        idx = 1;
    elseif Nslowopts == 2
        if iOpt == 1; idx = find(vp0_type == 1,1); else; idx = find(vp0_type == 2 | vp0_type == 3,1); end
    else
        idx = find(vp0_type == (mod(iOpt-1,3)+1),1);
    end
    starts(iOpt) = rescale_params(vp0_vec(idx)); %#ok<AGROW>
    vp0_type(idx) = []; vp0_vec(idx) = [];
    Theta0(:,iOpt) = get_vptheta(starts(iOpt)); %#ok<AGROW>
end
T = size(Theta0,1);

% (3) all chains in one on-device Adam loop per group of equal non-optimised parameters (one group in a default run)
master_stepsize.min = min(options.SGDStepSize,0.001);
if optimState.Warmup || ~vp.optimize_weights
    scaling_factor = min(0.1,options.SGDStepSize*10);
else
    scaling_factor = min(0.1,options.SGDStepSize);
end
master_stepsize.max = max(master_stepsize.min,scaling_factor);
master_stepsize.decay = 200;
MaxIter = min(options.MaxIterStochastic,1e4);
grp = fixed_groups(starts);
h = vbmc_hip_gp_handle(gp);
ThetaOpt = zeros(T,Nslowopts);
ThetaMid = zeros(T,Nslowopts);
for g = 1:max(grp)
    members = find(grp == g);
    [x,~,iters,xtab,ftab] = vbmc_hip_mex('adam',h,Theta0(:,members),starts(members(1)),NSentK,double(compute_var), ...
        elcbo_beta,thetabnd,randi(2^31-1),options.TolFunStochastic,MaxIter, ...
        [master_stepsize.min master_stepsize.max master_stepsize.decay]);
    ThetaOpt(:,members) = x;
    for r = 1:numel(members)
        [~,idx_mid] = min(ftab(1:double(iters(r)),r));      % best midpoint of the chain
        ThetaMid(:,members(r)) = xtab(:,idx_mid,r);
    end
end

% (4) full ELCBO at midpoints and endpoints: slots 2*iOpt-1 (midpoint, only with ELCBOmidpoint) and 2*iOpt (endpoint)
Nsgp = numel(gp.post);
NSentFineK = ceil(evaloption_vbmc(options.NSentFine,K)/K);
computevar_flag = ~(isfield(options,'SkipELBOVariance') && options.SkipELBOVariance);
nslot = 2*Nslowopts;
st.nelbo = Inf(1,nslot); st.nelcbo = Inf(1,nslot);
st.G = NaN(1,nslot); st.H = NaN(1,nslot); st.varF = NaN(1,nslot); st.varss = NaN(1,nslot);
st.theta = NaN(nslot,T); st.I_sk = NaN(nslot,Nsgp,K); st.J_sjk = NaN(nslot,Nsgp,K,K);
slot = []; chain = []; Th = zeros(T,0);
for iOpt = 1:Nslowopts
    if options.ELCBOmidpoint
        slot(end+1) = 2*iOpt-1; chain(end+1) = iOpt; Th(:,end+1) = ThetaMid(:,iOpt); %#ok<AGROW>
    end
    slot(end+1) = 2*iOpt; chain(end+1) = iOpt; Th(:,end+1) = ThetaOpt(:,iOpt); %#ok<AGROW>
end
for g = 1:max(grp)
    sel = find(grp(chain) == g);
    [F,~,varG,G,H,varGss,I_sk,J_sjk] = vbmc_hip_mex('elbo_batch',h,Th(:,sel),starts(chain(sel(1))),NSentFineK,0, ...
        double(computevar_flag),0,[],randi(2^31-1),1,Nsgp);
    if ~computevar_flag; varG = zeros(size(F)); varGss = zeros(size(F)); J_sjk = zeros(Nsgp,K,K,numel(sel)); end
    for q = 1:numel(sel)
        s = slot(sel(q));
        st.nelbo(s) = F(q); st.G(s) = G(q); st.H(s) = H(q); st.varF(s) = varG(q); st.varss(s) = varGss(q);
        st.nelcbo(s) = F(q) + elcbo_beta*sqrt(varG(q));
        st.theta(s,:) = Th(:,sel(q))';
        st.I_sk(s,:,:) = I_sk(:,:,q);
        st.J_sjk(s,:,:,:) = J_sjk(:,:,:,q);
    end
end

% (5) best slot
[~,idx] = min(st.nelcbo);
elbo = -st.nelbo(idx);
elbo_sd = sqrt(st.varF(idx));
G = st.G(idx); H = st.H(idx); varss = st.varss(idx); varG = st.varF(idx); varH = 0;
I_sk = zeros(Nsgp,K); J_sjk = zeros(Nsgp,K,K);
I_sk(:,:) = st.I_sk(idx,:,:);
J_sjk(:,:,:) = st.J_sjk(idx,:,:,:);
vp = rescale_params(starts(ceil(idx/2)),st.theta(idx,:));
vp.temperature = optimState.temperature;

% (6) pruning of components with negligible weight: one full-ELCBO evaluation (the shimmed negelcbo_vbmc) per attempt
pruned = 0;
if vp.optimize_weights
    alreadychecked = false(1,vp.K);
    while any(vp.w < options.TolWeight & ~alreadychecked)
        vp_pruned = vp;
        cand = find(vp_pruned.w < options.TolWeight & ~alreadychecked);
        idx = cand(randi(numel(cand)));
        vp_pruned.w(idx) = [];
        if isfield(vp_pruned,'eta'); vp_pruned.eta(idx) = []; end
        vp_pruned.sigma(idx) = [];
        vp_pruned.mu(:,idx) = [];
        vp_pruned.K = vp_pruned.K - 1;
        [theta_pruned,vp_pruned] = get_vptheta(vp_pruned,vp_pruned.optimize_mu,vp_pruned.optimize_sigma, ...
            vp_pruned.optimize_lambda,vp_pruned.optimize_weights);
        NSp = ceil(evaloption_vbmc(options.NSentFine,vp_pruned.K)/vp_pruned.K);
        [nelbo_p,~,G_p,H_p,varF_p,~,varss_p,varG_p,varH_p] = negelcbo_vbmc(theta_pruned(:)',0,vp_pruned,gp,NSp,0,computevar_flag,0,[],0);
        elbo_pruned = -nelbo_p;
        elbo_pruned_sd = sqrt(varF_p);
        delta_elcbo = abs((elbo_pruned - options.ELCBOImproWeight*elbo_pruned_sd) - (elbo - options.ELCBOImproWeight*elbo_sd));
        PruningThreshold = options.TolImprovement*evaloption_vbmc(options.PruningThresholdMultiplier,K);
        if delta_elcbo < PruningThreshold
            vp = vp_pruned;
            elbo = elbo_pruned; elbo_sd = elbo_pruned_sd;
            G = G_p; H = H_p; varss = varss_p; varG = varG_p; varH = varH_p;
            pruned = pruned + 1;
            alreadychecked(idx) = [];
            I_sk(:,idx) = [];
            J_sjk(:,:,idx) = [];
        else
            alreadychecked(idx) = true;
        end
    end
end

vp.stats.elbo = elbo;
vp.stats.elbo_sd = elbo_sd;
vp.stats.elogjoint = G;
vp.stats.elogjoint_sd = sqrt(varG);
vp.stats.entropy = H;
vp.stats.entropy_sd = sqrt(varH);
vp.stats.stable = false;
vp.stats.I_sk = I_sk;
vp.stats.J_sjk = J_sjk;
end

function grp = fixed_groups(vps)
% chains whose NON-optimised parameter groups are equal share a batch (those groups are not part of theta)
key = cell(1,numel(vps));
for i = 1:numel(vps)
    v = vps(i); fx = [];
    if ~v.optimize_mu; fx = [fx; v.mu(:)]; end %#ok<AGROW>
    if ~v.optimize_sigma; fx = [fx; v.sigma(:)]; end %#ok<AGROW>
    if ~v.optimize_lambda; fx = [fx; v.lambda(:)]; end %#ok<AGROW>
    if ~v.optimize_weights; fx = [fx; v.w(:)]; end %#ok<AGROW>
    key{i} = sprintf('%.17g,',fx);
end
[~,~,grp] = unique(key,'stable');
grp = grp(:)';
end
