--[[
Drop-in for gp.models.gp_regressor (ardse + GaussianNoise_iso + constant mean, bots/bayesopt.lua:39-45)
backed by b7_gp_set_data / b7_gp_fit_hyp / b7_gp_predict.
Register:  bot7.models.gp_hip = require('bot7hip.models_gp_hip')
and select it with config.model.type = 'gp_hip' (bots/bayesopt.lua:31), or pass it as cache.model.
Mirrors bot7_amd/models/gp_regressor.py method for method (that file is what the tests drive).

Hyper vector layout (ours; the gp rock's parse_hypers layout is not in the reference tree):
    { lenscale_sq_1..d, amp, noise, mean }
Hyper sampling (config.sampler = 'slice', bots/bayesopt.lua:44): with config.sample = true the reference's OWN
bot7.samplers.slice (samplers/slice.lua:51-168) walks
    log p(theta | X, Y) = -NLL(theta) + log prior(theta),   theta = { log lenscale_sq, log amp, log noise, mean }
with a flat prior inside explicit bounds (config.bounds; the gp rock's priors are unknown); every density evaluation
is one b7_gp_fit_hyp (d + 3 numbers uploaded, the data stay on the device).  config.sample = false (default):
sample_hypers returns the current point estimate.
--]]
local ffi = require('ffi')
local hip = require('bot7hip.bot7hip_ffi')

local title  = 'bot7.models.gp_hip'
local parent = 'bot7.models.abstract'
local model, parent = torch.class(title, parent)

function model:__init(config)
  parent.__init(self)
  self.config = config or {}
  self.hyp    = nil
  self.nEvals = 0
end

function model:init(X_obs, Y_obs)                       -- bots/abstract.lua:147-149
  local d, n = X_obs:size(2), Y_obs:size(1)
  local amp  = (n > 1) and Y_obs:var() * (n - 1) / n or 1.0       -- population variance, as numpy.var
  if not (amp > 0) then amp = 1.0 end
  self.hyp = {lenscale_sq = torch.DoubleTensor(d):fill(d / 8), amp = amp,
              noise = self.config.noiseless and 0.0 or 1e-4 * amp, mean = Y_obs:mean()}
  return self.hyp
end

-- ---- theta <-> hyp, bounds -------------------------------------------------------------------------------------------
local function to_theta(h)
  local d = h.lenscale_sq:nElement()
  local t = torch.DoubleTensor(d + 3)
  t:narrow(1, 1, d):copy(h.lenscale_sq):log()
  t[d+1] = math.log(h.amp); t[d+2] = math.log(math.max(h.noise, 1e-300)); t[d+3] = h.mean
  return t
end
local function from_theta(t)
  local d = t:nElement() - 3
  return {lenscale_sq = t:narrow(1, 1, d):clone():exp(), amp = math.exp(t[d+1]), noise = math.exp(t[d+2]), mean = t[d+3]}
end
function model:bounds(X, Y)
  local b, d = self.config.bounds or {}, X:size(2)
  local n  = Y:size(1)
  local vy = (n > 1) and Y:var() * (n - 1) / n or 1.0
  if not (vy > 0) then vy = 1.0 end
  local lo, hi = torch.DoubleTensor(d + 3), torch.DoubleTensor(d + 3)
  lo:narrow(1, 1, d):fill(math.log(b.lenscale_sq_min or 1e-3 * d)); hi:narrow(1, 1, d):fill(math.log(b.lenscale_sq_max or 1e3 * d))
  lo[d+1] = math.log(b.amp_min   or 1e-3 * vy); hi[d+1] = math.log(b.amp_max   or 1e3 * vy)
  lo[d+2] = math.log(b.noise_min or 1e-8 * vy); hi[d+2] = math.log(b.noise_max or 1e0 * vy)
  lo[d+3] = b.mean_min or (Y:min() - 3 * math.sqrt(vy)); hi[d+3] = b.mean_max or (Y:max() + 3 * math.sqrt(vy))
  return lo, hi
end

-- ---- fits ---------------------------------------------------------------------------------------------------------------
-- The data go to the device once per (X_obs, Y_obs) pair: the sampler and the marginalisation loop refit the SAME
-- tensors under new hypers (bots/bayesopt.lua:68-78).
-- keyed on the CONTENT (a copy of what was uploaded: N x (d + c) doubles, nothing next to a fit): pointers get reused by
-- the allocator and sums collide (Y and -Y with zero sum, two responses swapped)
local function same_data(self, X, Y)
  local k = self._data
  return k ~= nil and k.x:isSameSizeAs(X) and k.y:isSameSizeAs(Y) and k.x:equal(X) and k.y:equal(Y)
end

-- (X_obs, Y_obs) resident on the device; uploads only when they are not the pair already there.  Returns Y as N x c.
function model:stage_data(X_obs, Y_obs)
  local X = hip.pin(X_obs)
  local Y = hip.pin(Y_obs:dim() == 1 and Y_obs:view(-1, 1) or Y_obs)
  if not same_data(self, X, Y) then
    if hip.group then   -- every member of the group refits the same observations
      hip.gcheck(hip.C.b7_group_gp_set_data(hip.group, hip.data(X), hip.data(Y), X:size(1), X:size(2), Y:size(2)))
    else
      hip.check(hip.C.b7_gp_set_data(hip.ctx, hip.data(X), hip.data(Y), X:size(1), X:size(2), Y:size(2)))   -- synchronous
    end
    self._data = {x = X:clone(), y = Y:clone()}
  end
  return Y
end

-- data and candidate grid resident (for b7_eval_nominate, lua/bots_bayesopt_hip.lua)
function model:stage(X_obs, Y_obs, X_hid)
  self:stage_data(X_obs, Y_obs)
  if not hip.is_resident(X_hid) then hip.upload_grid(X_hid) end
end

function model:fit(X_obs, Y_obs, hyp, want_nll)
  local hyp = hyp or self.hyp
  local Y = self:stage_data(X_obs, Y_obs)
  local ls = hip.pin(hyp.lenscale_sq)
  local h  = ffi.new('b7_hyp', {hip.data(ls), hyp.amp, hyp.noise, hyp.mean})
  local nll, jit, info = ffi.new('double[?]', Y:size(2)), ffi.new('double[1]'), ffi.new('int[1]')
  hip.check(hip.C.b7_gp_fit_hyp(hip.ctx, h, want_nll and nll or nil, jit, info))
  if jit[0] > 0 then   -- the reference's warning text, utils/math.lua:210-212
    print(string.format('Warning: utils.math.chol succeeded in factorizing the\ninput matrix after applying a jitter of %.2e', jit[0]))
  elseif jit[0] < 0 then   -- :185-186
    print('Warning: utils.math.chol failed to factorize the input matrix; returning chol(I)')
  end
  self.last_fit = {jitter = jit[0], info = info[0]}
  return nll[0]
end

-- the likelihood alone (what the slice sampler evaluates, samplers/slice.lua:118-164): the data stay on the device, no inverse
-- and no alpha are built, and for N <= 128 it is one workgroup of one launch (b7_gp_nll_batch with one hyper vector)
function model:nll(X_obs, Y_obs, hyp)
  local hyp = hyp or self.hyp
  if (Y_obs:dim() == 2 and Y_obs:size(2) > 1) then return self:fit(X_obs, Y_obs, hyp, true) end
  local h = hyp.lenscale_sq
  local theta = torch.cat(h, torch.DoubleTensor{hyp.amp, hyp.noise, hyp.mean}):view(1, -1)
  return self:nll_batch(X_obs, Y_obs, theta)[1]
end

-- B likelihoods at once (thetas: B x (d+3) rows in the hyper-vector layout above): one persistent launch for all of them,
-- the current fit stays as it is.  For multi-chain samplers and speculative step-out probes.
function model:nll_batch(X_obs, Y_obs, thetas)
  local Y = self:stage_data(X_obs, Y_obs)
  local B, d = thetas:size(1), thetas:size(2) - 3
  local ls   = thetas:narrow(2, 1, d):contiguous()
  local amp, noise, mean = thetas:select(2, d + 1):contiguous(), thetas:select(2, d + 2):contiguous(), thetas:select(2, d + 3):contiguous()
  local out  = torch.DoubleTensor(B)
  hip.check(hip.C.b7_gp_nll_batch(hip.ctx, B, torch.data(ls), torch.data(amp), torch.data(noise), torch.data(mean),
                                  torch.data(out), nil, nil))
  return out
end

-- -NLL on the device + flat prior inside the bounds (-inf outside): the density bot7.samplers.slice evaluates
function model:log_posterior(theta, X_obs, Y_obs)
  local theta = theta:view(-1)
  local lo, hi = self:bounds(X_obs, Y_obs)
  if theta:lt(lo):any() or theta:gt(hi):any() or theta:ne(theta):any() then return -math.huge end
  -- the point a slice update starts from is the point the previous update ended on, whose density was the last thing evaluated
  -- (samplers/slice.lua:106 after :134-164): the same vector under the same data is not sent to the device again
  self:stage_data(X_obs, Y_obs)
  local memo = self._density_memo
  if memo and memo.data == self._data and memo.theta:equal(theta) then return memo.value end
  self.nEvals = self.nEvals + 1
  local value = -self:nll(X_obs, Y_obs, from_theta(theta))
  self._density_memo = {data = self._data, theta = theta:clone(), value = value}
  return value
end

-- config.chains = C > 1: C chains of the reference's own slice sampler advance in lock step.  Each chain runs inside a
-- coroutine whose density function YIELDS the point it wants evaluated; when every live chain has yielded, all the
-- points go to the device as one b7_gp_nll_batch and each coroutine is resumed with its value.  The sampler's code
-- (samplers/slice.lua:51-168) runs unmodified.  Mirrors gp_regressor._lockstep_update of the Python harness.
function model:lockstep_update(X_obs, Y_obs)
  local C, d = #self.chains, X_obs:size(2)
  local lo, hi = self:bounds(X_obs, Y_obs)
  local co, waiting, nwait = {}, {}, 0
  local function density(t, _)
    local tv = t:view(-1)
    if tv:lt(lo):any() or tv:gt(hi):any() or tv:ne(tv):any() then return -math.huge end   -- flat prior: no device call
    return coroutine.yield(tv:clone())
  end
  local function step(c, value)
    local ok, res = coroutine.resume(co[c], value)
    if not ok then error(res) end
    if coroutine.status(co[c]) == 'dead' then self.chains[c] = res; waiting[c] = nil
    else waiting[c] = res; nwait = nwait + 1 end
  end
  for c = 1, C do
    co[c] = coroutine.create(function()
      return self.sampler.sample(density, self.chains[c]:view(1, -1), self.sopt, nil)[1]:clone()
    end)
    step(c, nil)
  end
  while nwait > 0 do
    local ids, thetas = {}, torch.DoubleTensor(nwait, d + 3)
    for c = 1, C do if waiting[c] then ids[#ids + 1] = c; thetas[#ids]:copy(waiting[c]) end end
    local hyps = thetas:clone()
    hyps:narrow(2, 1, d + 2):exp()                       -- theta = {log lenscale_sq, log amp, log noise, mean}
    local nll = self:nll_batch(X_obs, Y_obs, hyps)
    self.nEvals = self.nEvals + #ids
    nwait = 0
    for i, c in ipairs(ids) do step(c, -nll[i]) end
  end
end

function model:sample_hypers_chains(X_obs, Y_obs, state)
  local C = self.config.chains
  if not self.chains then
    local Samplers = require('bot7.samplers')
    self.sampler = Samplers[self.config.sampler or 'slice']()
    self.sopt    = self.sampler.configure(self.config.sampler_opt or {})
    self.sopt.width, self.sopt.nSamples = self.sopt.width or 0.5, 1
    local t0, lo, hi = to_theta(self.hyp), self:bounds(X_obs, Y_obs)
    self.chains, self.pool = {}, {}
    for c = 1, C do   -- chain 1 starts at the point estimate, the others a little off it (inside the bounds)
      local t = t0:clone()
      if c > 1 then t:add(0.1, torch.randn(t:nElement())):cmax(lo):cmin(hi) end
      self.chains[c] = t
    end
  end
  if not state then
    for _ = 1, (self.config.nBurnin or 0) do self:lockstep_update(X_obs, Y_obs) end
    self.pool = {}
  else
    if #self.pool == 0 then
      self:lockstep_update(X_obs, Y_obs)
      for c = 1, C do self.pool[c] = self.chains[c]:clone() end
    end
    self.hyp = from_theta(table.remove(self.pool, 1))
  end
  local h = self.hyp
  return torch.cat(h.lenscale_sq, torch.DoubleTensor{h.amp, h.noise, h.mean})
end

function model:sample_hypers(X_obs, Y_obs, _, _, state) -- bots/bayesopt.lua:68 (burn-in) and :74 (state = true)
  if not self.hyp then self:init(X_obs, Y_obs) end
  if self.config.sample and (self.config.chains or 1) > 1 then return self:sample_hypers_chains(X_obs, Y_obs, state) end
  if self.config.sample then
    local Samplers = require('bot7.samplers')
    if not self.sampler then
      self.sampler = Samplers[self.config.sampler or 'slice']()          -- bots/bayesopt.lua:44
      self.sopt    = self.sampler.configure(self.config.sampler_opt or {})  -- samplers/slice.lua:32-48
      self.sopt.width = self.sopt.width or 0.5
    end
    if self.config.noiseless and not (self.hyp.noise > 0) then
      local lo = self:bounds(X_obs, Y_obs)
      self.hyp.noise = math.exp(lo[X_obs:size(2) + 2])
    end
    -- the chain's state is theta itself: as long as nobody replaced self.hyp, the next update starts from the very vector the
    -- last one returned (not from log(exp(theta))), where the density is known (log_posterior's memo)
    local kept      = self._chain_state
    local theta     = (kept and kept.hyp == self.hyp) and kept.theta or to_theta(self.hyp)
    local n_updates = state and 1 or (self.config.nBurnin or 0)
    local f = function(t, _) return self:log_posterior(t, X_obs, Y_obs) end
    self.sopt.nSamples = 1
    for _ = 1, n_updates do
      theta = self.sampler.sample(f, theta:view(1, -1), self.sopt, nil)[1]  -- samplers/slice.lua:51-89
    end
    self.hyp = from_theta(theta)
    self._chain_state = {theta = theta:clone(), hyp = self.hyp}
  end
  local h = self.hyp
  return torch.cat(h.lenscale_sq, torch.DoubleTensor{h.amp, h.noise, h.mean})
end

function model:parse_hypers(v)                          -- bots/bayesopt.lua:75
  local d = v:nElement() - 3
  return {lenscale_sq = v:narrow(1, 1, d):clone(), amp = v[d+1], noise = v[d+2], mean = v[d+3]}
end

-- ---- posterior ----------------------------------------------------------------------------------------------------------
-- fit + predict leaving mean/var on the device (used by the *_hip scores)
function model:predict_device(X_obs, Y_obs, X_hid, hyp)
  local hyp = hyp or self.hyp
  if hip.group then   -- M.ctx holds one shard: per-sample scoring over "the grid" has no meaning on it
    error('bot7hip: with a group of GPUs the scores run through bot7.bots.bayesopt_hip (b7_group_eval_nominate)')
  end
  if not hip.is_resident(X_hid) then hip.upload_grid(X_hid) end
  local Y = self:stage_data(X_obs, Y_obs)
  local ls = hip.pin(hyp.lenscale_sq)
  local h  = ffi.new('b7_hyp', {hip.data(ls), hyp.amp, hyp.noise, hyp.mean})
  local jit, info = ffi.new('double[1]'), ffi.new('int[1]')
  -- fit + posterior in one call: the prediction is enqueued behind the fit before the host has seen the pivot report
  hip.check(hip.C.b7_gp_predict_hyp(hip.ctx, h, nil, nil, nil, jit, info))
  self.last_fit = {jitter = jit[0], info = info[0]}
end

function model:predict(X_obs, Y_obs, X_hid, hyp, req)   -- scores/expected_improvement.lua:63
  local req  = req or {mean = true, var = true}
  local M    = X_hid:size(1)
  local c    = (Y_obs:dim() == 1) and 1 or Y_obs:size(2)
  local mean, var = torch.DoubleTensor(M, c), torch.DoubleTensor(M)
  self:fit(X_obs, Y_obs, hyp or self.hyp, false)
  if hip.is_resident_on_ctx(X_hid) then
    hip.check(hip.C.b7_gp_predict(hip.ctx, torch.data(mean), torch.data(var)))
  else
    local X = hip.pin(X_hid)
    hip.check(hip.C.b7_gp_predict_at(hip.ctx, hip.data(X), M, torch.data(mean), torch.data(var)))
  end
  return {mean = req.mean and mean or nil, var = req.var and var or nil}
end

function model:fantasize(nFantasies, X_obs, Y_obs, X_pend, hyp)   -- scores/expected_improvement.lua:57
  self:fit(X_obs, Y_obs, hyp or self.hyp, false)
  self.fcalls = (self.fcalls or 0) + 1
  local Xp  = hip.pin(X_pend)
  local P   = Xp:size(1)
  local out = torch.DoubleTensor(P, nFantasies)
  hip.check(hip.C.b7_gp_fantasize(hip.ctx, hip.data(Xp), P, nFantasies, (self.config.seed or 0) * 1000003 + self.fcalls,
                                  torch.data(out), nil, nil))
  return out
end

return model
