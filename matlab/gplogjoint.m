function [F,dF,varF,dvarF,varss,I_sk,J_sjk] = gplogjoint(vp,gp,grad_flags,avg_flag,jacobian_flag,compute_var,separate_K)
%GPLOGJOINT Drop-in shim: expected log joint (Bayesian quadrature) on an MI355X through vbmc_hip_mex.
%
% Same signature and defaulting as the reference (misc/gplogjoint.m:1-30).  Accelerated: (a) averaged over
% hyper-parameter samples (AVG_FLAG = 1) with Jacobian-transformed or (JACOBIAN_FLAG = 0) untransformed gradients for exactly the
% parameter groups VP optimises; (b) per-hyper-sample values (AVG_FLAG = 0: F and VARF are 1-by-Ns, DF is T-by-Ns, VARSS = 0) --
% the forms private/activesample_vbmc.m:155 ([~,~,varF] = gplogjoint(vp,gp,0,0,0,1)) and
% misc/vpoptimizeweights_vbmc.m:42 ([~,~,~,~,~,I_sk,J_sjk] = gplogjoint(vp,gp,0,0,0,1,1)) use; (c) DVARF, the gradient of the
% diagonal variance (COMPUTE_VAR = 2, with the Jacobians, averaged form); (d) per-component outputs together with gradients (two
% passes).  Other call forms go to the reference further down the path.
if nargin < 3; grad_flags = []; end
if nargin < 4 || isempty(avg_flag); avg_flag = true; end
if nargin < 5 || isempty(jacobian_flag); jacobian_flag = true; end
if nargin < 6; compute_var = []; end
if nargin < 7 || isempty(separate_K); separate_K = nargout > 5; end
if isempty(compute_var); compute_var = nargout > 2; end
if nargout < 2; grad_flags = false; elseif isempty(grad_flags); grad_flags = true; end
if isscalar(grad_flags); grad_flags = ones(1,4)*grad_flags; end
compute_vargrad = nargout > 3 && compute_var && any(grad_flags);
if compute_vargrad && compute_var ~= 2
    error('gplogjoint:FullVarianceGradient', ...
        'Computation of gradient of log joint variance is currently available only for diagonal approximation of the variance.');
end

vpflags = [vp.optimize_mu, vp.optimize_sigma, vp.optimize_lambda, vp.optimize_weights];
% shapes and model ids through vbmc_hip_supported (D, K, N, covariance, mean function, integrated mean, warping: the library's own
% limits), the call form here; whatever the library still refuses (vbmc_hip:unsupported) falls through below as well -- this
% function draws no random numbers, so a second pass on the reference is always exact
supported = vbmc_hip_supported(gp,vp,compute_var ~= 0) ...         % (round 5: dvarF without the Jacobians and per hyper-sample are accelerated too)
    && (~any(grad_flags) || isequal(logical(grad_flags(:)'),logical(vpflags))) ...
    && (~vp.optimize_weights || isfield(vp,'eta'));
if supported
    try
        [F,dF,varF,dvarF,varss,I_sk,J_sjk] = on_device(vp,gp,grad_flags,avg_flag,jacobian_flag,compute_var,separate_K,compute_vargrad);
        return;
    catch err
        if ~strcmp(err.identifier,'vbmc_hip:unsupported'); rethrow(err); end
    end
end
ref = vbmc_hip_reference('gplogjoint');
outs = cell(1,max(nargout,1));
[outs{:}] = ref(vp,gp,grad_flags,avg_flag,jacobian_flag,compute_var,separate_K);
outs(end+1:7) = {[]};
[F,dF,varF,dvarF,varss,I_sk,J_sjk] = outs{:};
end

function [F,dF,varF,dvarF,varss,I_sk,J_sjk] = on_device(vp,gp,grad_flags,avg_flag,jacobian_flag,compute_var,separate_K,compute_vargrad)
[theta,vp] = get_vptheta(vp);                  % misc/get_vptheta.m: the rescaled vp, so that theta and the fixed groups agree
h = vbmc_hip_gp_handle(gp);
g = any(grad_flags);
sepK2 = separate_K && g;               % the objective's entry point refuses the combination (negelcbo_vbmc.m:57-59): a pass of its own
if sepK2; separate_K = false; end
if avg_flag || numel(gp.post) == 1
    dvarF = [];
    if compute_vargrad
        [~,~,F,~,varF,~,varss,I_sk,J_sjk,dF,~,~,dvarF] = vbmc_hip_mex('elbo',h,theta(:),vp,0,1, ...
            2,double(separate_K),0,[],[],0,numel(gp.post),double(~jacobian_flag));
    else
        [~,~,F,~,varF,~,varss,I_sk,J_sjk,dF] = vbmc_hip_mex('elbo',h,theta(:),vp,0,double(g), ...
            double(compute_var),double(separate_K),0,[],[],0,numel(gp.post),double(~jacobian_flag));
    end
    if ~avg_flag; varss = 0; end
else                                            % misc/gplogjoint.m:398-399: no averaging, varss stays 0
    dvarF = [];
    if compute_vargrad                          % per-hyper-sample variance gradient, T x S (misc/gplogjoint.m:375-396, :407-409 skipped)
        [~,~,~,~,~,~,~,I_sk,J_sjk,~,F,varF,~,dF,dvarF] = vbmc_hip_mex('elbo',h,theta(:),vp,0,1, ...
            2,0,0,[],[],0,numel(gp.post),double(~jacobian_flag));
    elseif g
        [~,~,~,~,~,~,~,I_sk,J_sjk,~,F,varF,~,dF] = vbmc_hip_mex('elbo',h,theta(:),vp,0,1, ...
            double(compute_var),0,0,[],[],0,numel(gp.post),double(~jacobian_flag));
    else
        [~,~,~,~,~,~,~,I_sk,J_sjk,~,F,varF] = vbmc_hip_mex('elbo',h,theta(:),vp,0,0, ...
            double(compute_var),double(separate_K),0,[],[],0,numel(gp.post));
    end
    varss = 0;
end
if sepK2
    [~,~,~,~,~,~,~,I_sk,J_sjk] = vbmc_hip_mex('elbo',h,theta(:),vp,0,0,double(compute_var),1,0,[],[],0,numel(gp.post));
end
if ~g; dF = []; end
if ~compute_var; varF = []; varss = []; end
end
