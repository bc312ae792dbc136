function [vp0_vec,vp0_type,elcbo_beta,compute_var,NSentK,NSentKFast] = vpsieve_vbmc(Ninit,Nbest,vp,gp,optimState,options,K)
%VPSIEVE_VBMC Drop-in shim: the preliminary 'sieve' of variational-posterior candidates with ALL candidates
% evaluated in one batched pass on an MI355X (vbmc_hip_mex 'elbo_batch').
%
% Same signature as the reference (misc/vpsieve_vbmc.m:1).  Record and replay: the REFERENCE's own vpsieve_vbmc runs
% (found further down the path) and does all the bookkeeping -- sample counts, confidence weight, soft bounds, the
% candidates of vbinit_vbmc, the repository of earlier solutions -- while the negelcbo_vbmc shim of this directory only
% RECORDS each candidate it is asked to evaluate and answers 0 (vbmc_hip_state).  The reference's stable ascending sort of
% an all-zero vector leaves the candidates in their order of creation; the recorded batch is then evaluated in ONE device
% pass (vbmc_hip_sieve) and sorted here: index-identical to the sequential loop (:74-83), whatever version of the reference
% the user runs.  No line of the reference's sieve is restated here.
%
% VBMC_HIP_PARITY=1, or a surrogate / mixture outside the accelerated path (vbmc_hip_supported): the reference sieve
% runs unrecorded, candidate by candidate through the negelcbo_vbmc shim, which then draws exactly what the reference
% draws (or falls through itself).
if nargin < 7; K = []; end
ref = vbmc_hip_reference('vpsieve_vbmc');
probe = vp;
if ~isempty(K); probe.K = K; end
if isfield(optimState,'delta'); probe.delta = optimState.delta; else; probe.delta = 0; end
Neff = size(gp.X,1);
if isfield(optimState,'Neff'); Neff = optimState.Neff; end
need_var = evaloption_vbmc(options.ELCBOWeight,Neff) ~= 0;             % the sieve evaluates the variance iff the weight is not 0
if vbmc_hip_state('parity') || vbmc_hip_state('recording') || ~vbmc_hip_supported(gp,probe,need_var)
    [vp0_vec,vp0_type,elcbo_beta,compute_var,NSentK,NSentKFast] = ref(Ninit,Nbest,vp,gp,optimState,options,K);
    return;
end

vbmc_hip_state('record_begin');
guard = onCleanup(@() vbmc_hip_state('record_end'));                   % also when the reference raises
[vp0_vec,vp0_type,elcbo_beta,compute_var,NSentK,NSentKFast] = ref(Ninit,Nbest,vp,gp,optimState,options,K);
calls = vbmc_hip_state('record_end');
if isempty(calls); return; end                                         % Ninit = 0: nothing to evaluate
if numel(calls) ~= numel(vp0_vec)
    error('vbmc_hip:sieve','vpsieve_vbmc: %d candidates but %d recorded evaluations.',numel(vp0_vec),numel(calls));
end
score = vbmc_hip_sieve(calls,gp,elcbo_beta);
[~,order] = sort(score(:),'ascend');                                   % stable, NaN last
vp0_vec = vp0_vec(order);
vp0_type = vp0_type(order);
end
