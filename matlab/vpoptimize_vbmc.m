function [vp,varss,pruned] = vpoptimize_vbmc(Nfastopts,Nslowopts,vp,gp,K,optimState,options,prnt)
%VPOPTIMIZE_VBMC Drop-in shim: optimisation of the variational posterior with the Nslowopts Adam chains run in
% lock-step ENTIRELY on an MI355X (vbmc_hip_mex 'adam') and their 2*Nslowopts full-ELCBO evaluations in one batched pass.
%
% Same signature and defaulting as the reference (misc/vpoptimize_vbmc.m:1).  Written from the batched host mirror
% vbmc_amd/optimize.py:vpoptimize_vbmc (tested against a sequential restatement of the reference on the device's own
% random streams, tests/test_gpu_vpoptimize.py):
%   (1) batched sieve (matlab/vpsieve_vbmc.m), (2) one starting point per chain by candidate type, (3) all chains through
%   ONE on-device Adam loop (utils/fminadam.m per chain, including the 20-iteration stopping test), (4) one batched
%   evaluation of the full ELCBO (fine Monte Carlo entropy, full variance, per-component terms) at every chain's best
%   midpoint and endpoint, (5) the best of them, (6) component pruning (matlab/vbmc_hip_prune.m), (7) vp.stats.
%
% The whole call goes to the reference further down the path, BEFORE any random number is drawn, when
%   * VBMC_HIP_PARITY=1 -- the reference's own control flow then runs with only the leaf evaluations on the device (the
%     negelcbo_vbmc shim fed with MATLAB's randn blocks, the reference's host-loop fminadam), so the global random stream
%     is consumed exactly as in an unmodified run;
%   * the call is outside this path: deterministic entropy (NSentK = 0: fminunc), ELCBOWeight ~= 0 or StochasticOptimizer
%     other than 'adam' (CMA-ES), unsupported surrogates (vbmc_hip_supported), no options.
if nargin < 8; prnt = []; if nargin < 7; options = []; if nargin < 6; optimState = []; if nargin < 5; K = []; end; end; end; end
if ~inside_path(vp,gp,K,optimState,options)
    ref = vbmc_hip_reference('vpoptimize_vbmc');
    [vp,varss,pruned] = ref(Nfastopts,Nslowopts,vp,gp,K,optimState,options,prnt);
    return;
end
if isempty(K); K = vp.K; end
warm = ~vp.optimize_weights;
if isfield(optimState,'Warmup'); warm = optimState.Warmup; end
temperature = 1;
if isfield(optimState,'temperature'); temperature = optimState.temperature; end

% (1) the sieve: candidates in ascending order of their quick ELCBO estimate, with their types
[cand,ctype,elcbo_beta,compute_var,NSentK] = vpsieve_vbmc(Nfastopts,Nslowopts,vp,gp,optimState,options,K);
[vp,bnd] = vpbounds(vp,gp,options,K);                                 % soft bounds of the optimisation (misc/vpbounds.m)

% (2) chain c starts from the best candidate not taken yet whose type suits it: any type for a single chain; type 1, then
% "not type 1" for two chains; types 1,2,3,1,... for more (the selection of :53-61 with a mask instead of deletions)
taken = false(1,numel(cand));
ctype = ctype(:)';
want = mod((1:Nslowopts)-1,3) + 1;
one = Nslowopts == 1; two = Nslowopts == 2;
for c = 1:Nslowopts
    suits = one | (two & c == 1 & ctype == 1) | (two & c == 2 & (ctype == 2 | ctype == 3)) | (~one & ~two & ctype == want(c));
    pick = find(suits & ~taken,1);
    if isempty(pick); error('vbmc_hip:starts','vpoptimize_vbmc: no sieve candidate of the type chain %d starts from.',c); end
    taken(pick) = true;
    start(c) = rescale_params(cand(pick)); %#ok<AGROW>
    Theta0(:,c) = get_vptheta(start(c)); %#ok<AGROW>
end
T = size(Theta0,1);

% (3) the Adam loops on the device, one batch per group of chains whose NON-optimised parameters agree (one group in a
% default run).  Step sizes as :108-125: floor min(SGDStepSize,1e-3); ceiling SGDStepSize (x10 during warm-up), at most 0.1.
lo = min(options.SGDStepSize,0.001);
if warm || ~vp.optimize_weights; hi = options.SGDStepSize*10; else; hi = options.SGDStepSize; end
steps = [lo, max(lo,min(0.1,hi)), 200];
maxit = min(1e4,options.MaxIterStochastic);
grp = groups_of(start);
h = vbmc_hip_gp_handle(gp);
ThetaEnd = zeros(T,Nslowopts);
ThetaMid = zeros(T,Nslowopts);
for g = 1:max(grp)
    m = find(grp == g);
    try
        % xmid: each chain's iterate of smallest recorded objective (the best midpoint, :133), picked inside the library -- the
        % T x MaxIter x R iterate table never crosses into MATLAB
        [x,~,~,xmid] = vbmc_hip_mex('adam',h,Theta0(:,m),start(m(1)),NSentK,double(compute_var),elcbo_beta, ...
            bnd,randi(2^31-1),options.TolFunStochastic,maxit,steps);
        ThetaEnd(:,m) = x;
        ThetaMid(:,m) = xmid;
    catch err
        if ~strcmp(err.identifier,'vbmc_hip:unsupported'); rethrow(err); end
        % refused on the device: the user's own host-loop fminadam over the shimmed objective (which falls through by itself)
        ms = struct('min',steps(1),'max',steps(2),'decay',steps(3));
        for c = m
            fun = @(t) negelcbo_vbmc(t,elcbo_beta,start(c),gp,NSentK,1,compute_var,0,bnd,0);
            [xo,~,xl,fl] = fminadam(fun,Theta0(:,c)',[],[],options.TolFunStochastic,maxit,ms);
            ThetaEnd(:,c) = xo(:);
            [~,at] = min(fl);
            ThetaMid(:,c) = xl(at,:)';
        end
    end
end

% (4) full ELCBO at midpoints (slot 2c-1, only with ELCBOmidpoint) and endpoints (slot 2c).  Empty slots keep nelcbo = Inf
% (:33); a point with a non-finite parameter (a diverged chain) is not sent -- the library validates the whole batch -- and
% keeps NaN, which is what the reference's arithmetic gives it and what its min() skips.
S = numel(gp.post);
Nfine = ceil(evaloption_vbmc(options.NSentFine,K)/K);
with_variance = ~(isfield(options,'SkipELBOVariance') && options.SkipELBOVariance);
nslot = 2*Nslowopts;
res = struct('nelbo',Inf(1,nslot),'nelcbo',Inf(1,nslot),'G',NaN(1,nslot),'H',NaN(1,nslot),'varF',NaN(1,nslot), ...
    'varss',NaN(1,nslot),'theta',NaN(T,nslot),'I_sk',NaN(S,K,nslot),'J_sjk',NaN(S,K,K,nslot));
slot = zeros(1,0); chain = zeros(1,0);
for c = 1:Nslowopts
    if options.ELCBOmidpoint
        slot(end+1) = 2*c-1; chain(end+1) = c; res.theta(:,2*c-1) = ThetaMid(:,c); %#ok<AGROW>
    end
    slot(end+1) = 2*c; chain(end+1) = c; res.theta(:,2*c) = ThetaEnd(:,c); %#ok<AGROW>
end
for g = 1:max(grp)
    sel = find(grp(chain) == g);
    bad = ~all(isfinite(res.theta(:,slot(sel))),1);
    res.nelbo(slot(sel(bad))) = NaN; res.nelcbo(slot(sel(bad))) = NaN;
    sel = sel(~bad);
    if isempty(sel); continue; end
    vpg = start(chain(sel(1)));
    try
        [F,~,varG,G,H,varGss,I_sk,J_sjk] = vbmc_hip_mex('elbo_batch',h,res.theta(:,slot(sel)),vpg,Nfine,0, ...
            double(with_variance),0,[],randi(2^31-1),1,S);
        if ~with_variance; varG = zeros(size(F)); varGss = zeros(size(F)); J_sjk = zeros(S,K,K,numel(sel)); end
        for q = 1:numel(sel)
            s = slot(sel(q));
            res.nelbo(s) = F(q); res.G(s) = G(q); res.H(s) = H(q); res.varF(s) = varG(q); res.varss(s) = varGss(q);
            res.I_sk(:,:,s) = I_sk(:,:,q);
            res.J_sjk(:,:,:,s) = J_sjk(:,:,:,q);
        end
    catch err
        if ~strcmp(err.identifier,'vbmc_hip:unsupported'); rethrow(err); end
        for q = 1:numel(sel)                      % slot by slot through the shimmed objective (falls through on its own)
            s = slot(sel(q));
            [nF,~,G1,H1,vF,~,vss,~,~,I1,J1] = negelcbo_vbmc(res.theta(:,s)',0,vpg,gp,Nfine,0,with_variance,0,[],0);
            res.nelbo(s) = nF; res.G(s) = G1; res.H(s) = H1; res.varF(s) = vF; res.varss(s) = vss;
            res.I_sk(:,:,s) = I1;
            if isempty(J1); res.J_sjk(:,:,:,s) = 0; else; res.J_sjk(:,:,:,s) = J1; end
        end
    end
    res.nelcbo(slot(sel)) = res.nelbo(slot(sel)) + elcbo_beta*sqrt(res.varF(slot(sel)));
end

% (5) the best of them (first minimum; NaN skipped)
[~,b] = min(res.nelcbo);
best = struct('elbo',-res.nelbo(b),'elbo_sd',sqrt(res.varF(b)),'G',res.G(b),'H',res.H(b),'varss',res.varss(b), ...
    'varG',res.varF(b),'varH',0,'I_sk',res.I_sk(:,:,b),'J_sjk',res.J_sjk(:,:,:,b));
vp = rescale_params(start(ceil(b/2)),res.theta(:,b)');
vp.temperature = temperature;

% (6) pruning
[vp,best,pruned] = vbmc_hip_prune(vp,gp,options,K,best,with_variance);
varss = best.varss;

% (7) the statistics of the solution, fields in the reference's order (:244-252)
names = {'elbo','elbo_sd','elogjoint','elogjoint_sd','entropy','entropy_sd','stable','I_sk','J_sjk'};
vals = {best.elbo,best.elbo_sd,best.G,sqrt(best.varG),best.H,sqrt(best.varH),false,best.I_sk,best.J_sjk};
for q = 1:numel(names); vp.stats.(names{q}) = vals{q}; end
end

function ok = inside_path(vp,gp,K,optimState,options)
ok = false;
if isempty(options) || vbmc_hip_state('parity'); return; end
if isempty(K); K = vp.K; end
probe = vp; probe.K = K; probe.delta = 0;
if isfield(optimState,'delta'); probe.delta = optimState.delta; end
Neff = size(gp.X,1);
if isfield(optimState,'Neff'); Neff = optimState.Neff; end
deterministic = K == 1 || (isfield(optimState,'EntropySwitch') && optimState.EntropySwitch) ...
    || ceil(evaloption_vbmc(options.NSent,K)/K) == 0;
with_variance = ~(isfield(options,'SkipELBOVariance') && options.SkipELBOVariance);
ok = ~deterministic && evaloption_vbmc(options.ELCBOWeight,Neff) == 0 && strcmpi(options.StochasticOptimizer,'adam') ...
    && vbmc_hip_supported(gp,probe,with_variance);
end

function grp = groups_of(vps)
% chains whose NON-optimised parameter groups are equal share a batch (those groups are not part of theta)
label = cell(1,numel(vps));
for i = 1:numel(vps)
    v = vps(i); fx = [];
    if ~v.optimize_mu; fx = [fx; v.mu(:)]; end %#ok<AGROW>
    if ~v.optimize_sigma; fx = [fx; v.sigma(:)]; end %#ok<AGROW>
    if ~v.optimize_lambda; fx = [fx; v.lambda(:)]; end %#ok<AGROW>
    if ~v.optimize_weights; fx = [fx; v.w(:)]; end %#ok<AGROW>
    label{i} = sprintf('%.17g,',fx);
end
[~,~,grp] = unique(label,'stable');
grp = grp(:)';
end
