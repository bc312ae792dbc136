function [nlZ,dnlZ,post,K_mat,Q] = gplite_nlZ(hyp,gp,hprior)
%GPLITE_NLZ Drop-in shim: GP negative log marginal likelihood (+ gradient) on an MI355X through vbmc_hip_mex.
%
% Same signature as the reference (gplite/gplite_nlZ.m:1-28).  Accelerated: one or two outputs, SE-ARD
% covariance, mean function 0/1/4, no integrated mean, no output warping.  Requests for POST / K_MAT / Q and
% unsupported models go to the reference further down the path.  Extension: HYP with B > 1 columns returns
% NLZ (1 x B) and DNLZ (Nhyp x B) from one batched device pass (the reference raises
% gplite_nlZ:NoSampling for that form when a gradient is requested, :41-44).
if nargin < 3; hprior = []; end
supported = nargout <= 2 && gp.covfun(1) == 1 && any(gp.meanfun == [0 1 4]) ...
    && ~(isfield(gp,'intmeanfun') && gp.intmeanfun > 0) && ~(isfield(gp,'outwarpfun') && ~isempty(gp.outwarpfun));
if ~supported
    ref = vbmc_hip_reference('gplite_nlZ');
    outs = cell(1,max(nargout,1));
    [outs{:}] = ref(hyp,gp,hprior);
    outs(end+1:5) = {[]};
    [nlZ,dnlZ,post,K_mat,Q] = outs{:};
    return;
end
Nhyp = size(hyp,1);
if Nhyp ~= gp.Ncov+gp.Nnoise+gp.Nmean
    error('gplite_nlZ:dimmismatch','Number of hyperparameters mismatched with dimension of training inputs.');
end
if nargout > 1
    [nlZ,dnlZ] = vbmc_hip_mex('gp_nlz',hyp,gp.X,gp.y,gp.s2,gp.meanfun,gp.noisefun);
else
    nlZ = vbmc_hip_mex('gp_nlz',hyp,gp.X,gp.y,gp.s2,gp.meanfun,gp.noisefun);
end
if ~isempty(hprior)                     % gplite_nlZ.m:58-68
    for b = 1:size(hyp,2)
        if nargout > 1
            [P,dP] = gplite_hypprior(hyp(:,b),hprior);
            dnlZ(:,b) = dnlZ(:,b) - dP;
        else
            P = gplite_hypprior(hyp(:,b),hprior);
        end
        nlZ(b) = nlZ(b) - P;
    end
end
post = []; K_mat = []; Q = [];
end
