--[[
Drop-ins for bot7.scores.expected_improvement / confidence_bound (scores/*.lua) backed by b7_score_*.
Register:  local S = require('bot7hip.scores_hip')
           bot7.scores.expected_improvement = S.expected_improvement ; bot7.scores.confidence_bound = S.confidence_bound
They return the M-element score tensor like the originals (bots/bayesopt.lua:76 adds it); with a gp_hip model the
posterior never leaves the GPU between predict and score.  Mirrors bot7_amd/scores/*.py.
--]]
local ffi = require('ffi')
local hip = require('bot7hip.bot7hip_ffi')
local S   = {}

local function finish(M)
  local out = torch.DoubleTensor(M)
  local v, i = ffi.new('double[1]'), ffi.new('int64_t[1]')
  hip.check(hip.C.b7_score_finish(hip.ctx, 1.0, v, i, torch.data(out)))
  return out
end

do
  local EI, parent = torch.class('bot7.scores.expected_improvement_hip', 'bot7.scores.abstract')
  function EI:__init(config)
    parent.__init(self)
    local config = config or {}
    config['tradeoff']   = config.tradeoff or 0.0     -- scores/expected_improvement.lua:30
    config['nFantasies'] = config.nFantasies or 100   -- :31
    self.config = config
  end
  function EI:__call__(model, hyp, X_obs, Y_obs, X_hid, X_pend, config)
    local hyp, config = hyp or model.hyp, config or self.config
    local X_obs, Y_obs = X_obs, Y_obs
    if Y_obs:dim() == 1 then Y_obs = Y_obs:view(-1, 1) end
    -- fantasize outcomes for pending jobs and append them to the _obs tensors (:51-60)
    if torch.isTensor(X_pend) and X_pend:dim() > 0 and X_pend:size(1) > 0 then
      if X_pend:dim() == 1 then X_pend = X_pend:view(1, -1) end
      local Y_pend = model:fantasize(config.nFantasies, X_obs, Y_obs, X_pend, hyp)   -- nPend x nFantasies, :57
      X_obs = X_obs:cat(X_pend, 1)                                                     -- :58
      Y_obs = Y_obs:narrow(2, 1, 1):repeatTensor(1, config.nFantasies):cat(Y_pend, 1)  -- :59
    end
    model:predict_device(X_obs, Y_obs, X_hid, hyp)                                   -- :63
    local fmins = hip.pin(Y_obs:min(1):view(-1))                                     -- :64
    hip.check(hip.C.b7_score_reset(hip.ctx))
    hip.check(hip.C.b7_score_ei(hip.ctx, hip.data(fmins), config.tradeoff or 0.0))   -- :69-88 (row mean when > 1 column)
    return finish(X_hid:size(1))
  end
  S.expected_improvement = EI
end

do
  local CB, parent = torch.class('bot7.scores.confidence_bound_hip', 'bot7.scores.abstract')
  function CB:__init(config)
    parent.__init(self)
    local config = config or {}
    config['tradeoff']   = config.tradeoff or 1.0     -- scores/confidence_bound.lua:31-34
    config['nFantasies'] = config.nFantasies or 100
    config['bound']      = config.bound or 'lower'
    config['sign']       = config.sign or -1.0
    self.config = config
  end
  -- the reference's fantasies block re-declares its locals (scores/confidence_bound.lua:56): pending points have no
  -- effect there, and none here
  function CB:__call__(model, hyp, X_obs, Y_obs, X_hid, X_pend, config)
    local hyp, config = hyp or model.hyp, config or self.config
    model:predict_device(X_obs, Y_obs, X_hid, hyp)                                 -- :63
    hip.check(hip.C.b7_score_reset(hip.ctx))
    hip.check(hip.C.b7_score_cb(hip.ctx, config.tradeoff or 1.0,
                                (config.bound:lower() == 'upper') and 1 or 0, config.sign or -1.0))  -- :70-94
    return finish(X_hid:size(1))
  end
  S.confidence_bound = CB
end

return S
