function [H,dH] = entmc_vbmc(vp,Ns,grad_flags,jacobian_flag)
%ENTMC_VBMC Drop-in shim: Monte Carlo entropy of the variational posterior on an MI355X.
%
% Same signature and defaulting as the reference (ent/entmc_vbmc.m:1-14).  The device path evaluates the
% entropy term alone (vbmc_elbo_batch with a NULL surrogate, include/vbmc_hip.h) and returns the gradient for
% the flagged groups in the order [mu(:); log sigma; log lambda; eta] with the Jacobians applied
% (:110-125), or without them for JACOBIAN_FLAG = 0 (gradients with respect to sigma, lambda and w themselves).
% VBMC_HIP_PARITY=1: the K blocks randn(D,1,Ns/2) are drawn here in the reference's order (:53), and nothing else is drawn.
% A mixture beyond the library's limits (vbmc_hip_supported(): max_K, max_D) goes to the reference before any draw.
if nargin < 2 || isempty(Ns); Ns = 10; end
if nargout < 2; grad_flags = false; elseif nargin < 3 || isempty(grad_flags); grad_flags = true; end
if isscalar(grad_flags); grad_flags = ones(1,4)*grad_flags; end
if nargin < 4 || isempty(jacobian_flag); jacobian_flag = true; end
g = any(grad_flags);
lim = vbmc_hip_supported();
if vp.K > lim.max_K || vp.D > lim.max_D
    ref = vbmc_hip_reference('entmc_vbmc');
    if nargout > 1; [H,dH] = ref(vp,Ns,grad_flags,jacobian_flag); else; H = ref(vp,Ns,grad_flags,jacobian_flag); end
    return;
end
vpt = vp;
if g
    vpt.optimize_mu = logical(grad_flags(1)); vpt.optimize_sigma = logical(grad_flags(2));
    vpt.optimize_lambda = logical(grad_flags(3)); vpt.optimize_weights = logical(grad_flags(4));
end
[theta,vpt] = get_vptheta(vpt);                % misc/get_vptheta.m: the rescaled vp, so that theta and the fixed groups agree
% random numbers: exactly the reference's K blocks (:53) in parity mode and nothing else; one randi for the device stream otherwise
epsblk = []; seed = 0;
if vbmc_hip_state('parity')
    Nse = ceil(Ns/2)*2;
    epsblk = zeros(vp.D,Nse/2,vp.K);
    for j = 1:vp.K; epsblk(:,:,j) = reshape(randn(vp.D,1,Nse/2),[vp.D,Nse/2]); end
else
    seed = randi(2^31-1);
end
try
    [~,~,~,H,~,dH] = vbmc_hip_mex('elbo',uint64(0),theta(:),vpt,Ns,double(g),0,0,0,[],epsblk,seed,1,double(~jacobian_flag));
catch err
    if ~strcmp(err.identifier,'vbmc_hip:unsupported') || ~isempty(epsblk); rethrow(err); end    % after parity draws: no second pass
    ref = vbmc_hip_reference('entmc_vbmc');
    if nargout > 1; [H,dH] = ref(vp,Ns,grad_flags,jacobian_flag); else; H = ref(vp,Ns,grad_flags,jacobian_flag); end
    return;
end
if ~g; dH = []; end
end
