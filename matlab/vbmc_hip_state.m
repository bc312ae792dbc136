function out = vbmc_hip_state(cmd,arg)
%VBMC_HIP_STATE Process-wide switches of the shim layer.
%   vbmc_hip_state('parity')        true if VBMC_HIP_PARITY=1 is set in the environment: every shim then keeps the
%                                   reference's control flow and draws (only the leaf evaluations run on the device, fed
%                                   with MATLAB's own randn blocks), so that a vbmc() run consumes the global random stream
%                                   exactly as the unmodified reference does.
%   vbmc_hip_state('record_begin')  from here on the negelcbo_vbmc shim RECORDS its arguments and returns zeros instead of
%                                   evaluating (matlab/vpsieve_vbmc.m: the reference's own sieve builds the candidates,
%                                   the batch is evaluated afterwards in one device pass).
%   vbmc_hip_state('recording')     true between record_begin and record_end.
%   vbmc_hip_state('record_push',c) append one recorded call (a struct).
%   calls = vbmc_hip_state('record_end')   stop recording, return the cell array of recorded calls.
persistent active calls
out = [];
switch cmd
    case 'parity'
        out = strcmp(getenv('VBMC_HIP_PARITY'),'1');
    case 'record_begin'
        active = true; calls = {};
    case 'recording'
        out = ~isempty(active) && active;
    case 'record_push'
        calls{end+1} = arg;
    case 'record_end'
        out = calls; active = false; calls = {};
    otherwise
        error('vbmc_hip:usage','vbmc_hip_state: unknown command %s.',cmd);
end
end
