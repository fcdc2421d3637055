--[[
Drop-in for gp.models.gp_regressor (ardse + GaussianNoise_iso + constant mean, bots/bayesopt.lua:39-45)
backed by b7_gp_fit / b7_gp_predict.  Register:  bot7.models.gp_hip = require('bot7hip.models_gp_hip')
and select it with config.model.type = 'gp_hip' (bots/bayesopt.lua:31), or pass it as cache.model.

Hyper vector layout (ours): { lenscale_sq_1..d, amp, noise, mean }.
--]]
local ffi = require('ffi')
local hip = require('bot7hip.bot7hip_ffi')

local title  = 'bot7.models.gp_hip'
local parent = 'bot7.models.abstract'
local model, parent = torch.class(title, parent)

function model:__init(config)
  parent.__init(self)
  self.config = config or {}
  self.hyp = nil
end

function model:init(X_obs, Y_obs)                       -- bots/abstract.lua:147-149
  local d   = X_obs:size(2)
  local amp = (Y_obs:size(1) > 1) and Y_obs:var() * (Y_obs:size(1) - 1) / Y_obs:size(1) or 1.0
  if not (amp > 0) then amp = 1.0 end
  self.hyp = {lenscale_sq = torch.DoubleTensor(d):fill(d / 8), amp = amp,
              noise = self.config.noiseless and 0.0 or 1e-4 * amp, mean = Y_obs:mean()}
  return self.hyp
end

function model:sample_hypers(X_obs, Y_obs, _, _, state) -- bots/bayesopt.lua:68,74 (point estimate; slice sampling
  if not self.hyp then self:init(X_obs, Y_obs) end      -- over model:nll is the 8f-1 'next' row)
  local h = self.hyp
  return torch.cat(h.lenscale_sq, torch.DoubleTensor{h.amp, h.noise, h.mean})
end

function model:parse_hypers(v)                          -- bots/bayesopt.lua:75
  local d = v:nElement() - 3
  return {lenscale_sq = v:narrow(1, 1, d):clone(), amp = v[d+1], noise = v[d+2], mean = v[d+3]}
end

local function fit(X_obs, Y_obs, hyp, want_nll)
  local h = ffi.new('b7_hyp', {hip.ptr(hyp.lenscale_sq), hyp.amp, hyp.noise, hyp.mean})
  local nll, jit, info = ffi.new('double[1]'), ffi.new('double[1]'), ffi.new('int[1]')
  local Y = Y_obs:dim() == 1 and Y_obs:view(-1, 1) or Y_obs
  hip.check(hip.C.b7_gp_fit(hip.ctx, hip.ptr(X_obs), hip.ptr(Y), X_obs:size(1), X_obs:size(2), Y:size(2), h,
                            want_nll and nll or nil, jit, info))
  if jit[0] > 0 then   -- the reference's warning text, utils/math.lua:210-212
    print(string.format('Warning: utils.math.chol succeeded in factorizing the\ninput matrix after applying a jitter of %.2e', jit[0]))
  end
  return nll[0]
end

function model:nll(X_obs, Y_obs, hyp) return fit(X_obs, Y_obs, hyp or self.hyp, true) end

local function is_resident(X1)
  local r = hip.resident
  return r and r.version == hip.grid_version and r.rows == X1:size(1) and r.ptr == torch.data(X1)
end

-- fit + predict leaving mean/var on the device (used by the *_hip scores)
function model:predict_device(X_obs, Y_obs, X_hid, hyp)
  fit(X_obs, Y_obs, hyp or self.hyp, false)
  if not is_resident(X_hid) then
    local X = X_hid:contiguous()
    hip.check(hip.C.b7_grid_upload(hip.ctx, torch.data(X), X:size(1), X:size(2)))
    hip.grid_version = hip.grid_version + 1
    hip.resident = {ptr = torch.data(X_hid), rows = X_hid:size(1), version = hip.grid_version}
  end
  hip.check(hip.C.b7_gp_predict(hip.ctx, nil, nil))
end

function model:predict(X_obs, Y_obs, X_hid, hyp, req)   -- scores/expected_improvement.lua:63
  local req  = req or {mean = true, var = true}
  local M    = X_hid:size(1)
  local mean, var = torch.DoubleTensor(M, 1), torch.DoubleTensor(M)
  fit(X_obs, Y_obs, hyp or self.hyp, false)
  if is_resident(X_hid) then
    hip.check(hip.C.b7_gp_predict(hip.ctx, torch.data(mean), torch.data(var)))
  else
    hip.check(hip.C.b7_gp_predict_at(hip.ctx, hip.ptr(X_hid), M, torch.data(mean), torch.data(var)))
  end
  return {mean = req.mean and mean or nil, var = req.var and var or nil}
end

function model:fantasize(nFantasies, X_obs, Y_obs, X_pend, hyp)   -- scores/expected_improvement.lua:57
  fit(X_obs, Y_obs, hyp or self.hyp, false)
  self.fcalls = (self.fcalls or 0) + 1
  local P   = X_pend:size(1)
  local out = torch.DoubleTensor(P, nFantasies)
  hip.check(hip.C.b7_gp_fantasize(hip.ctx, hip.ptr(X_pend), P, nFantasies, (self.config.seed or 0) * 1000003 + self.fcalls,
                                  torch.data(out), nil, nil))
  return out
end

return model
