function [H,dH,gammasum] = entlb_vbmc(vp,grad_flags,jacobian_flag)
%ENTLB_VBMC Drop-in shim: deterministic entropy lower bound of the variational posterior on an MI355X.
%
% Same signature and defaulting as the reference (ent/entlb_vbmc.m:1-17); JACOBIAN_FLAG = 0 gives the gradients with respect to
% sigma, lambda and w themselves (:132-143).  The third output (gammasum) is served by the reference further down the path.
if nargout < 2; grad_flags = false; elseif nargin < 2 || isempty(grad_flags); grad_flags = true; end
if isscalar(grad_flags); grad_flags = ones(1,4)*grad_flags; end
if nargin < 3 || isempty(jacobian_flag); jacobian_flag = true; end
g = any(grad_flags);
if nargout > 2
    ref = vbmc_hip_reference('entlb_vbmc');
    outs = cell(1,max(nargout,1));
    [outs{:}] = ref(vp,grad_flags,jacobian_flag);
    outs(end+1:3) = {[]};
    [H,dH,gammasum] = outs{:};
    return;
end
vpt = vp;
if g
    vpt.optimize_mu = logical(grad_flags(1)); vpt.optimize_sigma = logical(grad_flags(2));
    vpt.optimize_lambda = logical(grad_flags(3)); vpt.optimize_weights = logical(grad_flags(4));
end
[theta,vpt] = get_vptheta(vpt);                % misc/get_vptheta.m: the rescaled vp, so that theta and the fixed groups agree
[~,~,~,H,~,dH] = vbmc_hip_mex('elbo',uint64(0),theta(:),vpt,0,double(g),0,0,0,[],[],0,1,double(~jacobian_flag));
if ~g; dH = []; end
end
