function h = vbmc_hip_gp_handle(gp,arg)
%VBMC_HIP_GP_HANDLE Upload gp.post once per distinct GP; free the previous one.
% The objective handle closes over a constant gp for a whole vpoptimize_vbmc call
% (misc/vpoptimize_vbmc.m:71), so a one-entry cache suffices.  The key is a fingerprint of EVERYTHING the device copy
% is built from -- gp.X, every gp.post(s).hyp and every gp.post(s).alpha (alpha changes whenever y, s2 or a
% hyper-parameter does; L is a function of X, hyp and s2) -- as three differently weighted checksums, O(N*D + S*N) per
% call: a stale hit would return silently wrong numbers, so no field is sampled.
%   h  = VBMC_HIP_GP_HANDLE(GP)             the surrogate on device 0 (what every single-device command takes)
%   hs = VBMC_HIP_GP_HANDLE(GP,'all')       one replica per device of a multi-device session (VBMC_HIP_DEVICES=n or
%                                           vbmc_hip_mex('comm_open',n)): the handle vector of 'elbo_batch_multi'
%   VBMC_HIP_GP_HANDLE(GP,NEWHANDLE)        registers a device surrogate that already exists for GP (the rank-one
%                                           append of gplite_post builds it on device 0), so that the next call does not upload it again.
persistent key handles
k = fingerprint(gp);
if nargin > 1 && ~ischar(arg)
    if ~isempty(handles) && ~any(handles == arg); release(handles); end
    handles = arg; key = k;
    h = arg;
    return;
end
everywhere = nargin > 1;                                  % 'all'
ndev = 1;
if everywhere; ndev = vbmc_hip_mex('comm_size'); end
if isempty(key) || ~isequal(k,key) || numel(handles) < ndev
    release(handles);
    if ndev > 1
        handles = vbmc_hip_mex('gp_upload_all',gp);
    else
        handles = vbmc_hip_mex('gp_upload',gp);
    end
    key = k;
end
if everywhere; h = handles; else; h = handles(1); end
end

function release(hs)
% replicas belong to their own device's context: the gateway frees a set through the communicator
if numel(hs) > 1; vbmc_hip_mex('gp_free_all',hs); elseif numel(hs) == 1; vbmc_hip_mex('gp_free',hs); end
end

function k = fingerprint(gp)
S = numel(gp.post);
hyp = [gp.post.hyp];                      % Nhyp x S
alpha = [gp.post.alpha];                  % N x S
mult = [gp.post.sn2_mult];
v = [gp.X(:); hyp(:); alpha(:); mult(:)];
n = numel(v);
w = 1 + mod((1:n)'*0.6180339887498949,1); % fixed, position-dependent weights
k = [size(gp.X), S, gp.meanfun, double(gp.noisefun(:)'), sum(v), w'*v, sum(abs(v).*w.^2)];
end
