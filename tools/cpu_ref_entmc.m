function cpu_ref_entmc(D, K, Ns, budget_s)
% CPU_REF_ENTMC  Interpreted CPU baseline for bench.py (SURVEY 8d): Monte-Carlo entropy of a Gaussian-mixture variational
% posterior and its reparameterisation gradient, written from the FORMULAS of ent/entmc_vbmc.m:28-128 (not from its text),
% vectorised over samples the way the reference is.  Runs in MATLAB and in Octave.  Prints
%     ENTMC_SECONDS_PER_EVAL <median seconds of one value+gradient evaluation at Ns samples per component>
% for a synthetic mixture of the benchmark's shape (seeded; the numbers themselves are not compared with anything -- the
% oracle and the HIP path are pinned elsewhere -- only the time is used, and bench.py labels it "port, interpreted").
if nargin < 4, budget_s = 10; end
rand('seed', 1); randn('seed', 1);                      %#ok<RAND> (Octave and MATLAB both accept the legacy seeding)
mu = 1.5 * randn(D, K);
sigma = 0.3 * exp(0.2 * randn(1, K));
lambda = ones(D, 1);
eta = 0.3 * randn(1, K);
w = exp(eta) / sum(exp(eta));
M = 2 * ceil(Ns / 2);                                   % an even number of samples: antithetic pairs
times = [];
t_all = tic;
while isempty(times) || (toc(t_all) < budget_s && numel(times) < 7)
    t0 = tic;
    [H, dH] = entmc_once(D, K, M, mu, sigma, lambda, w, eta);   %#ok<ASGLU>
    times(end + 1) = toc(t0);                           %#ok<AGROW>
end
fprintf('ENTMC_SECONDS_PER_EVAL %.6g\n', median(times));
end

function [H, dH] = entmc_once(D, K, M, mu, sigma, lambda, w, eta)
% H = -sum_j w_j mean_i log q(x_ij),  x_ij = mu_j + sigma_j lambda .* eps_i,  eps antithetic;  q = sum_k w_k N(x; mu_k, sigma_k^2 diag(lambda^2))
nf = (2 * pi) ^ (-D / 2) / prod(lambda);
H = 0;
g_mu = zeros(D, K); g_sigma = zeros(1, K); g_lambda = zeros(D, 1); g_w = zeros(1, K);
for j = 1:K
    e = randn(D, M / 2);
    e = [e, -e];                                        % D x M
    x = bsxfun(@plus, bsxfun(@times, e, lambda * sigma(j)), mu(:, j));
    q = zeros(1, M);
    lsum = zeros(D, M);                                 % sum_k w_k N_k (x - mu_k) ./ (sigma_k lambda).^2
    Nk = zeros(K, M);
    for k = 1:K
        z = bsxfun(@rdivide, bsxfun(@minus, x, mu(:, k)), lambda * sigma(k));
        nk = nf * sigma(k) ^ (-D) * exp(-0.5 * sum(z .^ 2, 1));
        Nk(k, :) = nk;
        q = q + w(k) * nk;
        lsum = lsum + bsxfun(@times, z, w(k) * nk ./ 1) ./ (sigma(k) * lambda * ones(1, M));
    end
    lq = log(q);
    H = H - w(j) * mean(lq);
    r = bsxfun(@rdivide, lsum, q);                      % D x M: (d/dx) -log q, up to sign conventions of the estimator
    g_mu(:, j) = w(j) * mean(r, 2);
    g_sigma(j) = w(j) * mean(sum(r .* bsxfun(@times, e, lambda), 1));
    g_lambda = g_lambda + w(j) * sigma(j) * mean(r .* e, 2);
    g_w(j) = g_w(j) - mean(lq);
    g_w = g_w - w(j) * mean(bsxfun(@rdivide, Nk, q), 2)';
end
% Jacobians of the parameterisation: log sigma, log lambda, softmax weights (eta)
g_sigma = g_sigma .* sigma;
g_lambda = g_lambda .* lambda;
ew = exp(eta); sw = sum(ew);
Jw = -(ew' * ew) / sw ^ 2 + diag(ew / sw);
g_eta = (Jw * g_w')';
dH = [g_mu(:); g_sigma(:); g_lambda(:); g_eta(:)];
end
