function [ymu,ys2,fmu,fs2,lp] = gplite_pred(gp,Xstar,ystar,s2star,ssflag,nowarpflag)
%GPLITE_PRED Drop-in shim: GP prediction on an MI355X through vbmc_hip_mex.
%
% Same signature and defaulting as the reference (gplite/gplite_pred.m:1-9).  The accelerated path
% covers what VBMC's acquisition sweep uses (SE-ARD covariance, mean functions 0/1/4, Gaussian noise
% models, no output warping, no integrated mean); every other call form
% goes to the reference further down the path.
if nargin < 3; ystar = []; end
if nargin < 4; s2star = []; end
if nargin < 5 || isempty(ssflag); ssflag = false; end
if nargin < 6 || isempty(nowarpflag); nowarpflag = false; end

supported = any(gp.meanfun == [0 1 4]) && gp.covfun(1) == 1 ...
    && ~(isfield(gp,'intmeanfun') && gp.intmeanfun > 0) ...
    && ~(isfield(gp,'outwarpfun') && ~isempty(gp.outwarpfun) && ~nowarpflag) ...
    && ~isempty(gp.post(1).alpha);
if ~supported
    ref = vbmc_hip_reference('gplite_pred');
    outs = cell(1,max(nargout,1));
    [outs{:}] = ref(gp,Xstar,ystar,s2star,ssflag,nowarpflag);
    outs(end+1:5) = {[]};
    [ymu,ys2,fmu,fs2,lp] = outs{:};
    return;
end
Nstar = size(Xstar,1);
if ~isempty(ystar) && size(ystar,1) ~= Nstar
    error('gplite_pred:ydimmismatch','YSTAR should be empty or a column vector of NSTAR observations.');
end
if ~isempty(s2star) && size(s2star,1) ~= Nstar
    error('gplite_pred:s2dimmismatch','S2STAR should be empty or a column vector of NSTAR estimated variances.');
end
h = vbmc_hip_gp_handle(gp);
[ymu,ys2,fmu,fs2] = vbmc_hip_mex('gp_pred',h,Xstar,ystar,s2star,double(ssflag),numel(gp.post));
lp = [];
if ~isempty(ystar) && nargout > 4       % log predictive density per hyper-sample (gplite_pred.m:124-127), O(Nstar*Ns) here
    if ssflag || numel(gp.post) == 1
        ymu_s = ymu; ys2_s = ys2;
    else
        [ymu_s,ys2_s] = vbmc_hip_mex('gp_pred',h,Xstar,ystar,s2star,1,numel(gp.post));
    end
    lp = -0.5*bsxfun(@minus,ystar,ymu_s).^2./ys2_s - 0.5*log(2*pi*ys2_s);
end
end
