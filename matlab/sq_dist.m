function C = sq_dist(a, b)
%SQ_DIST Drop-in shim for utils/sq_dist.m:14-50 on the GPU (MFMA f64 contraction).
if nargin<1 || nargin>3 || nargout>1, error('Wrong number of arguments.'); end
if nargin < 2; b = []; end
C = vbmc_hip_mex('sq_dist',a,b);
end
