function acq = acqwrapper_vbmc(Xs,vp,gp,optimState,transpose_flag,acqFun,acqInfo)
%ACQWRAPPER_VBMC Drop-in shim: acquisition sweep on an MI355X through vbmc_hip_mex.
%
% Same signature as the reference (acq/acqwrapper_vbmc.m:1).  Accelerated: vp.delta = 0 and the density-based
% acquisition functions acqf_vbmc / acqflog_vbmc / acqus_vbmc / acqfsn2_vbmc and the importance-sampled acqviqr_vbmc /
% acqimiqr_vbmc -- GP prediction for every
% hyper-sample, fbar / vtot, vbmc_pdf and the acquisition value are one fused device pass.  The integer mapping
% (:8) and the hard-bound test in the original space (:49-51) stay here (they need warpvars_vbmc).  Everything
% else (acqeig, vp.delta > 0, unsupported GP models) goes to the reference down the path.
ids = {'acqf_vbmc','acqflog_vbmc','acqus_vbmc','acqfsn2_vbmc','acqviqr_vbmc','acqimiqr_vbmc'};
id = find(strcmp(func2str(acqFun),ids),1) - 1;
supported = ~isempty(id) && ~(isfield(vp,'delta') && ~isempty(vp.delta) && any(vp.delta > 0)) ...
    && gp.covfun(1) == 1 && any(gp.meanfun == [0 1 4]) && ~(isfield(gp,'intmeanfun') && gp.intmeanfun > 0) ...
    && ~(isfield(gp,'outwarpfun') && ~isempty(gp.outwarpfun)) && gp.noisefun(3) == 0;
if ~supported
    ref = vbmc_hip_reference('acqwrapper_vbmc');
    acq = ref(Xs,vp,gp,optimState,transpose_flag,acqFun,acqInfo);
    return;
end
if transpose_flag; Xs = Xs'; end
Xs = real2int_vbmc(Xs,vp.trinfo,optimState.integervars);                 % :8
h = vbmc_hip_gp_handle(gp);
if id >= 4      % importance-sampled IQR functions: optimState.ActiveImportanceSampling lives on the device
    his = vbmc_hip_is_handle(h,optimState.ActiveImportanceSampling,id == 4);
    acq = vbmc_hip_mex('acq_iqr',h,his,Xs,optimState.gplengthscale,gp.X_rescaled,gp.sn2new, ...
        double(optimState.VarianceRegularizedAcqFcn),optimState.TolGPVar);
elseif id == 3
    acq = vbmc_hip_mex('acq',h,Xs,id,vp,optimState.ymax,double(optimState.VarianceRegularizedAcqFcn), ...
        optimState.TolGPVar,optimState.gplengthscale,gp.X_rescaled,gp.sn2new);
else
    acq = vbmc_hip_mex('acq',h,Xs,id,vp,optimState.ymax,double(optimState.VarianceRegularizedAcqFcn),optimState.TolGPVar);
end
X_orig = warpvars_vbmc(Xs,'i',vp.trinfo);                                 % :49-51
idx = any(bsxfun(@lt,X_orig,optimState.LBeps_orig),2) | any(bsxfun(@gt,X_orig,optimState.UBeps_orig),2);
acq(idx) = Inf;
if transpose_flag; acq = acq'; end
end
