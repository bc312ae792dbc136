function h = vbmc_hip_gp_handle(gp,newhandle)
%VBMC_HIP_GP_HANDLE Upload gp.post once per distinct GP; free the previous one.
% The objective handle closes over a constant gp for a whole vpoptimize_vbmc call
% (misc/vpoptimize_vbmc.m:71), so a one-entry cache keyed on a cheap fingerprint suffices.
% VBMC_HIP_GP_HANDLE(GP,NEWHANDLE) registers a device surrogate that already exists for GP (the rank-one
% append of gplite_post builds it on the device), so that the next call does not upload it again.
persistent key handle
k = [size(gp.X), numel(gp.post), gp.post(1).alpha(1), gp.post(1).hyp(1), gp.post(end).alpha(end)];
if nargin > 1
    if ~isempty(handle) && handle ~= newhandle; vbmc_hip_mex('gp_free',handle); end
    handle = newhandle; key = k;
elseif isempty(key) || ~isequal(k,key)
    if ~isempty(handle); vbmc_hip_mex('gp_free',handle); end
    handle = vbmc_hip_mex('gp_upload',gp);
    key = k;
end
h = handle;
end
