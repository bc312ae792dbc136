function C = sq_dist(a, b)
%SQ_DIST Drop-in shim for utils/sq_dist.m:14-50: large problems on the GPU (MFMA f64 contraction), small ones stay with
% the reference.  A MEX call costs two PCIe transfers and a launch (tens of microseconds), which the many tiny sq_dist
% calls inside VBMC (a handful of points against a handful of points) never win back: below VBMC_HIP_SQDIST_MIN output
% elements (default 250000, e.g. 500 x 500) the reference further down the path is called instead.
persistent minel ref
if nargin<1 || nargin>3 || nargout>1, error('Wrong number of arguments.'); end
if nargin < 2; b = []; end
if isempty(minel)
    minel = str2double(getenv('VBMC_HIP_SQDIST_MIN'));
    if isnan(minel); minel = 250000; end
    ref = vbmc_hip_reference('sq_dist');
end
if isempty(b); m = size(a,2); else; m = size(b,2); end
if nargin > 2 || size(a,2)*m < minel || size(a,1) > 32      % third argument (Q matrix form) is not accelerated
    if nargin > 2; error('vbmc_hip:sq_dist','three-argument form not supported by the shim'); end
    if isempty(b); C = ref(a); else; C = ref(a,b); end
    return;
end
C = vbmc_hip_mex('sq_dist',a,b);
end
