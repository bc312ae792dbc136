function his = vbmc_hip_is_handle(h,ais,use_ctmp)
%VBMC_HIP_IS_HANDLE Upload optimState.ActiveImportanceSampling once per active-sampling step; free the previous one.
% The struct is rebuilt by activeimportancesampling_vbmc once per acquired point (private/activesample_vbmc.m:209-212)
% and then read by every acquisition call of that search (the 8192-point sweep and the CMA-ES refinement).
persistent key handle
k = [double(h), size(ais.Xa), ais.Xa(1), ais.Xa(end), ais.fs2a(1), ais.fs2a(end), sum(ais.lnw(:))];
if isempty(key) || ~isequal(k,key)
    if ~isempty(handle); vbmc_hip_mex('is_free',handle); end
    if use_ctmp && isfield(ais,'Ctmp_mat'); ct = ais.Ctmp_mat; else; ct = []; end   % IMIQR: solved on the device
    handle = vbmc_hip_mex('is_create',h,ais.Xa,ais.lnw,ais.fs2a,ct);
    key = k;
end
his = handle;
end
