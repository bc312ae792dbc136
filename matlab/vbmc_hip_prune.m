function [vp,best,nremoved] = vbmc_hip_prune(vp,gp,options,K,best,with_variance)
%VBMC_HIP_PRUNE Removal of mixture components of negligible weight after an optimisation (misc/vpoptimize_vbmc.m:196-243).
%   [VP,BEST,NREMOVED] = VBMC_HIP_PRUNE(VP,GP,OPTIONS,K,BEST,WITH_VARIANCE).  BEST holds the statistics of the current
%   solution (fields elbo, elbo_sd, G, H, varss, varG, varH, I_sk, J_sjk) and comes back updated; NREMOVED counts the removals.
%
% Written against the structure of vbmc_amd/optimize.py (the tested host mirror), with the surviving components tracked
% by their ORIGINAL index: `origin(j)` is the index component j had on entry, `kept` lists the original indices that were
% tried and had to stay.  Per attempt: one component among those below TolWeight and not yet tried, chosen by ONE randi (the
% reference's draw, :204); the mixture without it is evaluated by the full ELCBO (the negelcbo_vbmc shim: fine Monte
% Carlo entropy for K-1 components, full variance); it is dropped if the lower confidence bounds with and without it differ
% by less than TolImprovement*PruningThresholdMultiplier(K).  Attempts depend on each other, so they run one at a time.
% The per-component arrays lose column j of I_sk and slice (:,:,j) of J_sjk -- the third dimension only, as the
% reference does (:239), so that vp.stats is the reference's.
nremoved = 0;
if ~vp.optimize_weights; return; end
origin = 1:vp.K;
kept = [];
margin = options.TolImprovement*evaloption_vbmc(options.PruningThresholdMultiplier,K);
lcb = @(m,s) m - options.ELCBOImproWeight*s;
while true
    open = find(vp.w < options.TolWeight & ~ismember(origin,kept));
    if isempty(open); break; end
    j = open(randi(numel(open)));
    trial = without_component(vp,j);
    [theta_t,trial] = get_vptheta(trial,trial.optimize_mu,trial.optimize_sigma,trial.optimize_lambda,trial.optimize_weights);
    Nfine = ceil(evaloption_vbmc(options.NSentFine,trial.K)/trial.K);
    [nF,~,G,H,vF,~,vss,vG,vH] = negelcbo_vbmc(theta_t(:)',0,trial,gp,Nfine,0,with_variance,0,[],0);
    if abs(lcb(-nF,sqrt(vF)) - lcb(best.elbo,best.elbo_sd)) < margin
        vp = trial;
        best.elbo = -nF; best.elbo_sd = sqrt(vF);
        best.G = G; best.H = H; best.varss = vss; best.varG = vG; best.varH = vH;
        best.I_sk(:,j) = [];
        best.J_sjk(:,:,j) = [];
        origin(j) = [];
        nremoved = nremoved + 1;
    else
        kept(end+1) = origin(j); %#ok<AGROW>
    end
end
end

function v = without_component(v,j)
v.w(j) = [];
if isfield(v,'eta'); v.eta(j) = []; end
v.sigma(j) = [];
v.mu(:,j) = [];
v.K = v.K - 1;
end
