function ok = vbmc_hip_supported(gp,vp,need_var)
%VBMC_HIP_SUPPORTED True if the surrogate GP and the variational posterior are inside the accelerated path
% (the cases libvbmc_hip.so answers with VBMC_ERR_UNSUPPORTED otherwise; see INTEGRATION.md section 4), so that
% the shims can decide to fall through to the reference BEFORE they consume any random numbers.
%   vbmc_hip_supported(gp)              the surrogate alone (mean function 0/1/4, SE-ARD, no integrated mean / warping, D <= 32)
%   vbmc_hip_supported(gp,vp)           ... and the mixture (K <= 256, vp.delta = 0)
%   vbmc_hip_supported(gp,vp,need_var)  ... and, if NEED_VAR, what the variance path needs (N <= 3872, the factors gp.post(s).L)
% The limits are the ones elbo_plan enforces (vbmc_amd/csrc/abi_elbo.hip; DESIGN.md section 8 "Limits that remain").
D = size(gp.X,2);
ok = any(gp.meanfun == [0 1 4]) && gp.covfun(1) == 1 ...
    && ~(isfield(gp,'intmeanfun') && ~isempty(gp.intmeanfun) && gp.intmeanfun > 0) ...
    && ~(isfield(gp,'outwarpfun') && ~isempty(gp.outwarpfun)) ...
    && D <= 32 && ~isempty(gp.post) && ~isempty(gp.post(1).alpha);
if nargin > 1 && ok
    ok = vp.K <= 256 && ~(isfield(vp,'delta') && ~isempty(vp.delta) && any(vp.delta(:) ~= 0));
end
if nargin > 2 && ok && need_var
    ok = size(gp.X,1) <= 3872 && ~isempty(gp.post(1).L);
end
end
