function gp = gplite_post(hyp,X,y,covfun,meanfun,noisefun,s2,update1,outwarpfun)
%GPLITE_POST Drop-in shim: GP posterior (kernel matrix, jittered Cholesky, alpha) for all hyper-parameter
% samples in one batched device call through vbmc_hip_mex.
%
% Same signature and defaulting as the reference (gplite/gplite_post.m:1-29).  Accelerated: the
% from-scratch form GPLITE_POST(HYP,X,Y,COVFUN,MEANFUN,NOISEFUN,S2) with SE-ARD covariance, mean
% function 0/1/4, no integrated mean, no output warping; and the rank-one append
% GPLITE_POST(GP,XSTAR,YSTAR,[],[],[],[],1) of one noiseless-input observation (:173-251), which runs on the
% device and leaves the enlarged surrogate there.  The re-computation form and everything else are
% forwarded to the reference further down the path.
if nargin < 4; covfun = []; end
if nargin < 5; meanfun = []; end
if nargin < 6; noisefun = []; end
if nargin < 7; s2 = []; end
if nargin < 8 || isempty(update1); update1 = false; end
if nargin < 9; outwarpfun = []; end

if isstruct(hyp) && update1 && isempty(s2) && isempty(outwarpfun) && size(X,1) == 1 && rank1_supported(hyp)
    gp = hyp; xstar = X; ystar = y;
    Ns = numel(gp.post); sn2_eff = zeros(Ns,1);                    % [mstar,vstar] of :189 are formed inside the library
    for s = 1:Ns
        hn = gp.post(s).hyp(gp.Ncov+1:gp.Ncov+gp.Nnoise);
        sn2_eff(s) = gplite_noisefun(hn,xstar,gp.noisefun,ystar,[])*gp.post(s).sn2_mult;   % :205-207
    end
    Xn = [gp.X; xstar];
    [alpha,L,hnew] = vbmc_hip_mex('gp_rank1',vbmc_hip_gp_handle(gp),Xn,ystar,[],[],sn2_eff);
    for s = 1:Ns
        gp.post(s).alpha = alpha(:,s);
        gp.post(s).L = L(:,:,s);
        gp.post(s).sW = [gp.post(s).sW; 1/sqrt(sn2_eff(s))];     % :239
    end
    gp.X = Xn; gp.y = [gp.y; ystar];
    vbmc_hip_gp_handle(gp,hnew);                                   % the enlarged surrogate is already on the device
    return;
end

if isempty(covfun); cf = 1; else; cf = covfun; end
if isempty(meanfun); mf = 1; else; mf = meanfun; end
supported = ~isstruct(hyp) && ~update1 && isempty(outwarpfun) && ~iscell(mf) && isnumeric(mf) ...
    && any(mf(1) == [0 1 4]) && isnumeric(cf) && cf(1) == 1 && ~isempty(hyp) && ~isempty(y);
if ~supported
    ref = vbmc_hip_reference('gplite_post');
    args = {hyp,X,y,covfun,meanfun,noisefun,s2,update1,outwarpfun};
    gp = ref(args{1:max(nargin,1)});
    return;
end

% struct layout: gplite/gplite_post.m:94-157
gp.X = X; gp.y = y; gp.s2 = s2;
[Nhyp,Ns] = size(hyp);
if isempty(noisefun); if isempty(s2); noisefun = [1 0 0]; else; noisefun = [1 1 0]; end; end
[gp.Ncov,info] = gplite_covfun('info',X,cf);        gp.covfun = info.covfun;
[gp.Nnoise,info] = gplite_noisefun('info',X,noisefun); gp.noisefun = info.noisefun;
[gp.Nmean,info] = gplite_meanfun('info',X,mf,y);    gp.meanfun = info.meanfun; gp.meanfun_extras = info.extras;
gp.intmeanfun = 0; gp.intmeanfun_mean = []; gp.intmeanfun_var = [];
gp.outwarpfun = [];
if Nhyp ~= gp.Ncov+gp.Nnoise+gp.Nmean
    error('gplite_post:dimmismatch','Number of hyperparameters mismatched with GP model specification.');
end
[alpha,L,sW,mult,lch,hdev] = vbmc_hip_mex('gp_post',hyp,X,y,s2,gp.meanfun,gp.noisefun);   % hdev: the posterior the device just built
for s = 1:Ns
    gp.post(s).hyp = hyp(:,s);
    gp.post(s).alpha = alpha(:,s);
    gp.post(s).sW = sW(:,s);
    gp.post(s).L = L(:,:,s);
    gp.post(s).sn2_mult = mult(s);
    gp.post(s).Lchol = logical(lch(s));
end
% the factorisation stays where it was computed: registered for this gp, so that the next negelcbo_vbmc / gplogjoint / gplite_pred on
% it uploads nothing (through round 5 the gateway freed it here and the first evaluation sent X, alpha and all of L back:
% 25.6 MB at N = 400, S = 20).  gplite/gplite_post.m:167-172 is where the reference finishes the struct.
vbmc_hip_gp_handle(gp,hdev);
end

function ok = rank1_supported(gp)
% the configurations the device append covers; the reference itself takes the full update for the others (:76-90)
ok = gp.covfun(1) == 1 && isnumeric(gp.meanfun) && any(gp.meanfun(1) == [0 1 4]) && isempty(gp.s2) ...
    && ~(isfield(gp,'intmeanfun') && ~isempty(gp.intmeanfun) && gp.intmeanfun > 0) ...
    && ~(isfield(gp,'outwarpfun') && ~isempty(gp.outwarpfun));
end
