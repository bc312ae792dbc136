function ok = vbmc_hip_supported(gp,vp,need_var)
%VBMC_HIP_SUPPORTED True if the surrogate GP and the variational posterior are inside the accelerated path
% (the cases libvbmc_hip.so answers with VBMC_ERR_UNSUPPORTED otherwise; see INTEGRATION.md section 4), so that
% the shims can decide to fall through to the reference BEFORE they consume any random numbers.
%   vbmc_hip_supported(gp)              the surrogate alone (mean function, SE-ARD, no integrated mean / warping, D)
%   vbmc_hip_supported(gp,vp)           ... and the mixture (K; vp.delta where the library carries it)
%   vbmc_hip_supported(gp,vp,need_var)  ... and, if NEED_VAR, what the variance path needs (N, the factors gp.post(s).L)
%   lim = vbmc_hip_supported()          the limits themselves
% The numbers are the LIBRARY's: vbmc_hip_mex('limits') returns what its validation enforces (vbmc_get_limits, include/vbmc_hip.h:
% max_D, max_K, max_N, max_Na, max_T_vargrad, delta_ok, meanfun), asked once per session -- a shim that restates them goes stale
% (through round 5 this file said K <= 256 and N <= 3872 while the library took 512 and 10208).
persistent lim
if isempty(lim); lim = vbmc_hip_mex('limits'); end
if nargin == 0; ok = lim; return; end
D = size(gp.X,2);
ok = any(gp.meanfun == lim.meanfun) && gp.covfun(1) == 1 ...
    && ~(isfield(gp,'intmeanfun') && ~isempty(gp.intmeanfun) && gp.intmeanfun > 0) ...
    && ~(isfield(gp,'outwarpfun') && ~isempty(gp.outwarpfun)) ...
    && D <= lim.max_D && ~isempty(gp.post) && ~isempty(gp.post(1).alpha);
if nargin > 1 && ok
    ok = vp.K <= lim.max_K && (lim.delta_ok || ~(isfield(vp,'delta') && ~isempty(vp.delta) && any(vp.delta(:) ~= 0)));
end
if nargin > 2 && ok && need_var
    ok = size(gp.X,1) <= lim.max_N && ~isempty(gp.post(1).L);
end
end
