function dump_golden(repo_root, vbmc_root)
%DUMP_GOLDEN Evaluate the committed golden INPUTS with the real reference (acerbilab/vbmc, MATLAB).
%
%   dump_golden('/path/to/this/repo', '/path/to/vbmc')
%
% For every tests/golden/mp_case*.json, mp_nlz_case*.json, mp_pred_case*.json, mp_pen_case*.json and mp_acq_case*.json this runs the reference's own functions
% (gplite_post, gplite_pred, gplogjoint, entmc_vbmc, entlb_vbmc, gplite_nlZ, vpbndloss, negelcbo_vbmc, acqf/acqflog/acqus/acqfsn2/acqviqr_vbmc) on the stored
% inputs and writes tests/golden/matlab/matlab_case*.json / matlab_nlz_case*.json / matlab_pred_case*.json / matlab_pen_case*.json /
% matlab_acq_case*.json (mp_case4.json is BASELINE configs[0]: the GP trained on the reference's own rosenbrock_test target).
% tools/compare_matlab_golden.py then compares those files with the mpmath vectors (and thereby with the oracle and the HIP path,
% which are pinned to the mpmath vectors by the test-suite); COMMITTING the tests/golden/matlab/ folder turns the suite's
% "oracle pinned-by-MATLAB: absent" into "present" (tests/test_matlab_pin.py checks every entry on every run).
%
% The one command:   matlab -batch "dump_golden('/path/to/this/repo','/path/to/vbmc')" && python tools/compare_matlab_golden.py  Nothing here is needed by CI: the development container has no MATLAB,
% which is exactly why the oracle is documented as "parity unpinned by the reference" -- this script is how a
% maintainer WITH MATLAB closes that gap.
%
% The Monte-Carlo draws of entmc_vbmc come from the global randn stream (ent/entmc_vbmc.m:53).  To feed it the stored
% draws the script puts a temporary randn.m in front of the built-in that pops pre-loaded blocks (restored afterwards).
addpath(vbmc_root); vbmc('all');                                   % adds acq, ent, gplite, misc, shared, utils (vbmc.m:1056-1078)
gold = fullfile(repo_root,'tests','golden');
outdir = fullfile(gold,'matlab');
if ~exist(outdir,'dir'); mkdir(outdir); end
files = dir(fullfile(gold,'mp_case*.json'));
shadow = tempname; mkdir(shadow);
fid = fopen(fullfile(shadow,'randn.m'),'w');
fprintf(fid,'function r = randn(varargin)\nglobal DUMP_GOLDEN_QUEUE\nr = DUMP_GOLDEN_QUEUE{1}; DUMP_GOLDEN_QUEUE(1) = [];\nassert(isequal(size(r),[varargin{:}]));\nend\n');
fclose(fid);
cleanup = onCleanup(@() rmpath(shadow));
for f = 1:numel(files)
    rec = jsondecode(fileread(fullfile(gold,files(f).name)));
    in = rec.inputs; D = in.D; K = in.K; S = in.S; Mh = in.Mh;
    X = reshape_rows(in.X,D); y = in.y(:); hyp = reshape_rows(in.hyp,S);
    mu = reshape_rows(in.mu,K); eps3 = in.eps;                     % eps: K x Mh x D
    gp = gplite_post(hyp,X,y,1,in.meanfun);                        % covfun 1 (SE-ARD), noisefun default [1 0 0]
    vp = struct('D',D,'K',K,'mu',mu,'sigma',in.sigma(:)','lambda',in.lam(:),'w',in.w(:)','eta',in.eta(:)', ...
        'optimize_mu',true,'optimize_sigma',true,'optimize_lambda',true,'optimize_weights',true,'delta',[],'trinfo',[]);
    out = struct();
    % entropy (Monte Carlo, stored draws in the reference's order: K blocks randn(D,1,Mh))
    global DUMP_GOLDEN_QUEUE %#ok<TLEV>
    DUMP_GOLDEN_QUEUE = cell(1,K);
    for j = 1:K; DUMP_GOLDEN_QUEUE{j} = reshape(permute(eps3(j,:,:),[3 1 2]),[D,1,Mh]); end
    addpath(shadow,'-begin');
    [out.entmc_H,out.entmc_dH] = entmc_vbmc(vp,2*Mh,true,true);
    rmpath(shadow);
    [out.entlb_H,out.entlb_dH] = entlb_vbmc(vp,true,true);
    % GP posterior and prediction
    Xs = reshape_rows(in.Xstar,D);
    [~,~,fmu,fs2] = gplite_pred(gp,Xs,[],[],1,0);
    out.alpha = cell(1,S); out.L = cell(1,S);
    for s = 1:S; out.alpha{s} = gp.post(s).alpha; out.L{s} = gp.post(s).L; end
    out.pred_fmu = fmu'; out.pred_fs2 = fs2';
    % expected log joint per hyper-sample (avg_flag = 0), gradient, full and diagonal variance
    [Fs,dFs,varF1,~,~,I_sk,J_sjk] = gplogjoint(vp,gp,[1 1 1 1],0,1,1,1);  %#ok<ASGLU>  gradient of the VALUE only
    out.G_s = Fs(:)'; out.dG_s = dFs'; out.I_sk = I_sk; out.J_sjk = J_sjk; out.varG_s_full = varF1(:)';
    [~,~,varF2] = gplogjoint(vp,gp,[0 0 0 0],0,1,2,0);
    out.varG_s_diag = varF2(:)';
    write_json(fullfile(outdir,strrep(files(f).name,'mp_','matlab_')),out);
end
files = dir(fullfile(gold,'mp_nlz_case*.json'));
for f = 1:numel(files)
    rec = jsondecode(fileread(fullfile(gold,files(f).name)));
    in = rec.inputs; D = in.D; S = in.S;
    X = reshape_rows(in.X,D); y = in.y(:); hyp = reshape_rows(in.hyp,S);
    s2 = []; if ~isempty(in.s2); s2 = in.s2(:); end
    gp = gplite_post(hyp(:,1),X,y,1,in.meanfun,in.noisefun(:)',s2);
    out = struct('nlZ',zeros(1,S),'dnlZ',zeros(S,size(hyp,1)));
    for s = 1:S
        [out.nlZ(s),g] = gplite_nlZ(hyp(:,s),gp,[]);
        out.dnlZ(s,:) = g(:)';
    end
    write_json(fullfile(outdir,strrep(files(f).name,'mp_','matlab_')),out);
end
% prediction with the general noise models (gplite_noisefun.m:176-210), ystar / s2star and the log predictive density
files = dir(fullfile(gold,'mp_pred_case*.json'));
for f = 1:numel(files)
    rec = jsondecode(fileread(fullfile(gold,files(f).name)));
    in = rec.inputs; D = in.D; S = in.S;
    X = reshape_rows(in.X,D); y = in.y(:); hyp = reshape_rows(in.hyp,S);
    s2 = []; if ~isempty(in.s2); s2 = in.s2(:); end
    s2star = []; if ~isempty(in.s2star); s2star = in.s2star(:); end
    gp = gplite_post(hyp,X,y,1,in.meanfun,in.noisefun(:)',s2);
    Xs = reshape_rows(in.Xstar,D);
    [~,ys2,fmu,fs2,lp] = gplite_pred(gp,Xs,in.ystar(:),s2star,1,0);
    out = struct('fmu',fmu','fs2',fs2','ys2',ys2','lp',lp','alpha',zeros(S,size(X,1)),'min_sn2',zeros(1,S));
    for s = 1:S
        out.alpha(s,:) = gp.post(s).alpha';
        out.min_sn2(s) = 1/gp.post(s).sW(1)^2/gp.post(s).sn2_mult;     % gplite_core.m:281
    end
    write_json(fullfile(outdir,strrep(files(f).name,'mp_','matlab_')),out);
end
% soft-bound and weight penalties (misc/vpbndloss.m, utils/softbndloss.m, misc/negelcbo_vbmc.m:146-162): vpbndloss directly, the
% weight penalty as negelcbo_vbmc adds it (difference of the calls with and without thetabnd on a one-point surrogate, no entropy
% samples: everything else cancels)
files = dir(fullfile(gold,'mp_pen_case*.json'));
for f = 1:numel(files)
    rec = jsondecode(fileread(fullfile(gold,files(f).name)));
    in = rec.inputs; D = in.D; K = in.K; o = logical(in.opt(:)');
    vp = struct('D',D,'K',K,'mu',reshape_rows(in.mu,K),'sigma',in.sigma(:)','lambda',in.lam(:),'eta',in.eta(:)', ...
        'w',exp(in.eta(:)')/sum(exp(in.eta(:))),'optimize_mu',o(1),'optimize_sigma',o(2),'optimize_lambda',o(3),'optimize_weights',o(4), ...
        'delta',[],'trinfo',[]);
    tb = struct('lb',in.lb(:),'ub',in.ub(:),'TolCon',in.TolCon,'WeightThreshold',in.WeightThreshold,'WeightPenalty',in.WeightPenalty);
    theta = in.theta(:);
    [Lb,dLb] = vpbndloss(theta,vp,tb,tb.TolCon);
    gp1 = gplite_post([zeros(D,1); 0; log(1e-3); 0],zeros(1,D),0,1,1);      % one training point, constant mean
    [F1,dF1] = negelcbo_vbmc(theta,0,vp,gp1,0,1,0,0,tb);
    [F0,dF0] = negelcbo_vbmc(theta,0,vp,gp1,0,1,0,0,[]);
    out = struct('L_bnd',Lb,'dL_bnd',dLb(:)','L_w',(F1-F0)-Lb,'dL_w',(dF1(:)-dF0(:))'-dLb(:)');
    write_json(fullfile(outdir,strrep(files(f).name,'mp_','matlab_')),out);
end
% acquisition functions (acq/acqf_vbmc.m, acqflog_vbmc.m, acqus_vbmc.m, acqfsn2_vbmc.m, acqviqr_vbmc.m) on the stored points, called
% directly with the statistics acqwrapper_vbmc.m:17-29 forms (the wrapper itself needs vp.trinfo / warpvars_vbmc)
files = dir(fullfile(gold,'mp_acq_case*.json'));
for f = 1:numel(files)
    rec = jsondecode(fileread(fullfile(gold,files(f).name)));
    in = rec.inputs; D = in.D; K = in.K; S = in.S;
    X = reshape_rows(in.X,D); y = in.y(:); hyp = reshape_rows(in.hyp,S);
    gp = gplite_post(hyp,X,y,1,in.meanfun);
    gl = in.gplengthscale(:)';
    gp.X_rescaled = bsxfun(@rdivide,X,gl); gp.sn2new = in.sn2new(:);
    vp = struct('D',D,'K',K,'mu',reshape_rows(in.mu,K),'sigma',in.sigma(:)','lambda',in.lam(:),'w',in.w(:)','delta',[],'trinfo',[]);
    Xs = reshape_rows(in.Xstar,D); Xa = reshape_rows(in.Xa,D); Na = size(Xa,1); N = size(X,1);
    [~,~,fmu,fs2] = gplite_pred(gp,Xs,[],[],1,0);                                   % acqwrapper_vbmc.m:17
    fbar = sum(fmu,2)/S; vbar = sum(fs2,2)/S;                                       % :21-29
    if S > 1; vf = sum(bsxfun(@minus,fmu,fbar).^2,2)/(S-1); else; vf = 0; end
    vtot = vf + vbar;
    optimState = struct('ymax',in.ymax,'gplengthscale',gl,'VarianceRegularizedAcqFcn',false,'TolGPVar',1e-4);
    % importance-sampling state as private/activeimportancesampling_vbmc.m:248-276 leaves it (unit weights)
    AIS = struct('Xa',Xa,'lnw',zeros(S,Na),'Kax_mat',zeros(Na,N,S),'Ctmp_mat',zeros(N,Na,S));
    [~,~,~,fs2a] = gplite_pred(gp,Xa,[],[],1,0); AIS.fs2a = fs2a;
    for s = 1:S
        h = gp.post(s).hyp; ell = exp(h(1:D)); sf2 = exp(2*h(D+1)); L = gp.post(s).L;
        Kax = sf2*exp(-sq_dist(diag(1./ell)*Xa',diag(1./ell)*X')/2);
        AIS.Kax_mat(:,:,s) = Kax;
        if gp.post(s).Lchol; sn2_eff = 1/gp.post(s).sW(1)^2; AIS.Ctmp_mat(:,:,s) = (L\(L'\Kax'))/sn2_eff; else; AIS.Ctmp_mat(:,:,s) = L*Kax'; end
    end
    optimState.ActiveImportanceSampling = AIS;
    out = struct('fbar',fbar(:)','vtot',vtot(:)');
    out.acqf = acqf_vbmc(Xs,vp,gp,optimState,fmu,fs2,fbar,vtot)';
    out.acqflog = acqflog_vbmc(Xs,vp,gp,optimState,fmu,fs2,fbar,vtot)';
    out.acqus = acqus_vbmc(Xs,vp,gp,optimState,fmu,fs2,fbar,vtot)';
    out.acqfsn2 = acqfsn2_vbmc(Xs,vp,gp,optimState,fmu,fs2,fbar,vtot)';
    out.acqviqr = acqviqr_vbmc(Xs,vp,gp,optimState,fmu,fs2,fbar,vtot)';
    write_json(fullfile(outdir,strrep(files(f).name,'mp_','matlab_')),out);
end
end

function A = reshape_rows(v,ncol)
% jsondecode returns a matrix for rectangular nested lists (rows = outer list): make sure of the shape
A = v; if isvector(A) && ncol > 1 && numel(A) ~= ncol; A = A(:); end
if size(A,2) ~= ncol && size(A,1) == ncol; A = A'; end
end

function write_json(path,s)
fid = fopen(path,'w'); fwrite(fid,jsonencode(s)); fclose(fid);
fprintf('wrote %s\n',path);
end
