function h = vbmc_hip_gp_handle(gp)
%VBMC_HIP_GP_HANDLE Upload gp.post once per distinct GP; free the previous one.
% The objective handle closes over a constant gp for a whole vpoptimize_vbmc call
% (misc/vpoptimize_vbmc.m:71), so a one-entry cache keyed on a cheap fingerprint suffices.
persistent key handle
k = [size(gp.X), numel(gp.post), gp.post(1).alpha(1), gp.post(1).hyp(1), gp.post(end).alpha(end)];
if isempty(key) || ~isequal(k,key)
    if ~isempty(handle); vbmc_hip_mex('gp_free',handle); end
    handle = vbmc_hip_mex('gp_upload',gp);
    key = k;
end
h = handle;
end
