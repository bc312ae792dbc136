function [x,f,iters] = vbmc_hip_fminadam(Theta0,elcbo_beta,vp,gp,NSentK,compute_var,thetabnd,TolFun,MaxIter,master_stepsize)
%VBMC_HIP_FMINADAM The Adam loop of misc/vpoptimize_vbmc.m:127 for one or several chains, entirely on the device.
%
% Replaces
%     vpfun = @(theta_) negelcbo_vbmc(theta_,elcbo_beta,vp0,gp,NSentK,1,compute_var,0,thetabnd);
%     [thetaopt,~,theta_lst,fval_lst] = fminadam(vpfun,theta0,[],[],TolFunAdam,[],master_stepsize);
% by  thetaopt = vbmc_hip_fminadam(theta0(:),elcbo_beta,vp0,gp,NSentK,compute_var,thetabnd,TolFunAdam,[],master_stepsize);
% Columns of THETA0 are independent chains run in lock-step (utils/fminadam.m:42-102 per chain, including
% the 20-iteration slope / random-walk stopping test); X holds the mean of each chain's last 20 iterates.
if nargin < 8 || isempty(TolFun); TolFun = 0.001; end
if nargin < 9 || isempty(MaxIter); MaxIter = 1e4; end
step = [0.001 0.1 200];                              % fminadam.m:11-13
if nargin > 9 && ~isempty(master_stepsize)
    if isfield(master_stepsize,'min'); step(1) = master_stepsize.min; end
    if isfield(master_stepsize,'max'); step(2) = master_stepsize.max; end
    if isfield(master_stepsize,'decay'); step(3) = master_stepsize.decay; end
end
h = vbmc_hip_gp_handle(gp);
[x,f,iters] = vbmc_hip_mex('adam',h,Theta0,vp,NSentK,double(compute_var),elcbo_beta,thetabnd,randi(2^31-1), ...
    TolFun,MaxIter,step);
end
