function ok = vbmc_hip_supported(gp,vp)
%VBMC_HIP_SUPPORTED True if the surrogate GP and the variational posterior are inside the accelerated path
% (the cases libvbmc_hip.so answers with VBMC_ERR_UNSUPPORTED otherwise; see INTEGRATION.md section 4), so that
% the batched shims can decide to fall through to the reference BEFORE they consume any random numbers.
D = size(gp.X,2);
ok = any(gp.meanfun == [0 1 4]) && gp.covfun(1) == 1 ...
    && ~(isfield(gp,'intmeanfun') && ~isempty(gp.intmeanfun) && gp.intmeanfun > 0) ...
    && ~(isfield(gp,'outwarpfun') && ~isempty(gp.outwarpfun)) ...
    && D <= 32 && ~isempty(gp.post) && ~isempty(gp.post(1).alpha);
if nargin > 1 && ok
    K = vp.K;
    ok = K <= 256 && (4*D*K + 9*K <= 19400) && ~(isfield(vp,'delta') && ~isempty(vp.delta) && any(vp.delta(:) ~= 0));
end
end
