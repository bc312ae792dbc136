function [F,dF,G,H,varF,dH,varGss,varG,varH,I_sk,J_sjk] = negelcbo_vbmc(theta,beta,vp,gp,Ns,compute_grad,compute_var,altent_flag,thetabnd,entropy_alpha)
%NEGELCBO_VBMC Drop-in shim: negative ELCBO on an MI355X through vbmc_hip_mex.
%
% Same signature, nargin/nargout defaulting and error ids as the reference
% (misc/negelcbo_vbmc.m:1-24).  Put this directory BEFORE the VBMC folders on the path.  Anything
% outside the accelerated path (unsupported mean function, weights-only optimisation, a mixture or a
% training set beyond the library's limits -- vbmc_hip_supported asks the library for them -- ...) falls
% through to the reference implementation found further down the path.
%
% Random numbers.  The reference draws K blocks randn(D,1,Ns/2) per call when Ns > 0 (ent/entmc_vbmc.m:53) and nothing
% when Ns == 0 (entlb_vbmc, misc/negelcbo_vbmc.m:104-110).  VBMC_HIP_PARITY=1: exactly those blocks are drawn here, in
% that order, and handed to the device (eps_mode 1) -- no other draw, so MATLAB's global stream advances as in the
% reference and the result matches it to fp64 round-off.  Otherwise (device Philox stream, eps_mode 0) ONE randi seeds the
% device stream of a call with Ns > 0; a call with Ns == 0 draws nothing in either mode.
if nargin < 5 || isempty(Ns); Ns = 0; end
if nargin < 6 || isempty(compute_grad); compute_grad = nargout > 1; end
if nargin < 7; compute_var = []; end
if nargin < 8 || isempty(altent_flag); altent_flag = false; end %#ok<NASGU>
if nargin < 9; thetabnd = []; end
if nargin < 10 || isempty(entropy_alpha); entropy_alpha = 0; end %#ok<NASGU>
if isempty(beta) || ~isfinite(beta); beta = 0; end
if isempty(compute_var); compute_var = beta ~= 0 || nargout > 4; end
separate_K = nargout > 9;

if vbmc_hip_state('recording')      % matlab/vpsieve_vbmc.m: note the candidate, evaluate the whole batch later
    vbmc_hip_state('record_push',struct('theta',theta(:),'vp',vp,'Ns',Ns,'compute_var',compute_var,'thetabnd',thetabnd));
    F = 0; dF = []; G = 0; H = 0; varF = 0; dH = []; varGss = 0; varG = 0; varH = 0; I_sk = []; J_sjk = [];
    return;
end

onlyweights = vp.optimize_weights && ~vp.optimize_mu && ~vp.optimize_sigma && ~vp.optimize_lambda;
epsblk = []; seed = 0;
drawn = false;                                  % has this call consumed random numbers already?
try
    if onlyweights; error('vbmc_hip:unsupported','weights-only branch stays on the host'); end
    if ~vbmc_hip_supported(gp,vp,compute_var)   % decided from shapes and model ids, BEFORE any random number is drawn
        error('vbmc_hip:unsupported','outside the accelerated path');
    end
    h = vbmc_hip_gp_handle(gp);                 % cached upload, keyed on the gp struct (see INTEGRATION.md)
    if Ns > 0
        if vbmc_hip_state('parity')
            Nse = ceil(Ns/2)*2;
            epsblk = zeros(vp.D,Nse/2,vp.K);
            for j = 1:vp.K; epsblk(:,:,j) = reshape(randn(vp.D,1,Nse/2),[vp.D,Nse/2]); end
        else
            seed = randi(2^31-1);
        end
        drawn = true;
    end
    [F,dF,G,H,varG,dH,varGss,I_sk,J_sjk] = vbmc_hip_mex('elbo',h,theta(:),vp,Ns,double(compute_grad), ...
        double(compute_var),double(separate_K),beta,thetabnd,epsblk,seed,numel(gp.post));
    varH = 0;
    if compute_var; varF = varG + varH; else; varF = 0; varG = 0; varGss = 0; end
catch err
    if ~strcmp(err.identifier,'vbmc_hip:unsupported'); rethrow(err); end
    if drawn && vbmc_hip_state('parity')
        % the K blocks are consumed already: the reference would draw them again and leave the stream K blocks ahead
        error('vbmc_hip:parity','negelcbo_vbmc: the device refused a call after its random draws (%s); parity with the reference stream is lost.',err.message);
    end
    ref = vbmc_hip_reference('negelcbo_vbmc');  % next negelcbo_vbmc on the path (which -all)
    outs = cell(1,max(nargout,1));
    [outs{:}] = ref(theta,beta,vp,gp,Ns,compute_grad,compute_var,false,thetabnd,0);
    [F,dF,G,H,varF,dH,varGss,varG,varH,I_sk,J_sjk] = deal_padded(outs,11);
end
end

function varargout = deal_padded(c,n)
c(end+1:n) = {[]};
varargout = c(1:n);
end
