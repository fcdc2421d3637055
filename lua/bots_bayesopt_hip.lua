--[[
bot7.bots.bayesopt with eval + nominate as ONE library call (b7_eval_nominate).

bots/bayesopt.lua:56-99 draws nSamples hyper vectors, adds one acquisition vector per sample on the host, divides,
and takes score:max(1).  With the *_hip model and scores that is nSamples fits + posteriors + score:adds on the GPU
and one M-vector download per sample.  This subclass keeps the sampling exactly as the parent does it (:68 burn-in
call, :74-75 per sample) but hands all nSamples hyper tables to the library at once: the fits, posteriors and
score:adds are enqueued back to back, the arg-max follows on the device (across GPUs when dist_hip has set up a
communicator), and the host waits once.  Nothing but the winner's index comes back.

Register:  bot7.bots.bayesopt = require('bot7hip.bots_bayesopt_hip')      -- or pass it as the bot class
Falls back to the parent's nominate whenever the model or score is not a *_hip one (e.g. DNGO).

Several GPUs, two layouts (INTEGRATION.md section 4):
  * ONE process, a group of GPUs (bot7hip_ffi.use_group{...}): self.candidates stays the reference's host tensor of ALL
    candidates, nominate's index is an index into it, and the parent's run_trial (bots/abstract.lua:112-152) runs
    unchanged -- the steal hook deletes the row on whichever GPU holds it.
  * one process per GPU (dist_hip.init): self.candidates is THIS RANK'S SHARD, nominate's index is 1-based in the UNION of
    the shards, and run_trial below replaces the parent's line 118 by dist_hip.commit; everything else is the parent's
    code.  The random initial picks (bots/bayesopt.lua:90-91) are drawn against the union's row count.
The harness's Python stand-in for this file is harness/bots/bayesopt.py + harness/dist.py.
--]]
local ffi = require('ffi')
local hip = require('bot7hip.bot7hip_ffi')
local D   = require('bot7hip.dist_hip')

local title  = 'bot7.bots.bayesopt_hip'
local parent = 'bot7.bots.bayesopt'
local bot, parent = torch.class(title, parent)

function bot:__init(objective, hypers, config, cache)
  parent.__init(self, objective, hypers, config, cache)
end

-- b7_score_spec of a *_hip score object, or nil when the score has no device form
local function score_spec(score, Y_obs, keep)
  local cfg, kind = score.config or {}, torch.type(score)
  if kind == 'bot7.scores.expected_improvement_hip' then
    local fmins = hip.pin(Y_obs:min(1):view(-1))            -- scores/expected_improvement.lua:64
    keep[#keep + 1] = fmins
    return ffi.new('b7_score_spec', {hip.SCORE_EI, cfg.tradeoff or 0.0, 0, 0.0, hip.data(fmins)})
  elseif kind == 'bot7.scores.confidence_bound_hip' then
    local upper = (string.lower(cfg.bound or 'lower') == 'upper') and 1 or 0   -- scores/confidence_bound.lua:72
    return ffi.new('b7_score_spec', {hip.SCORE_CB, cfg.tradeoff or 1.0, upper, cfg.sign or -1.0, nil})
  end
  return nil
end

function bot:nominate(candidates)
  local candidates = candidates or self.candidates
  if self.nTrials <= self.config.bot.nInitial then                          -- bots/bayesopt.lua:90-91
    -- sharded one process per GPU: against the UNION's rows (every rank draws the same number from the same stream)
    local rows = (D.world > 1) and assert(D.M_global, 'dist_hip.shard_range first') or candidates:size(1)
    return torch.rand(1):mul(rows):long():add(1)
  end
  local model, keep = self.model, {}
  local Y_obs = self.responses
  if Y_obs:dim() == 1 then Y_obs = Y_obs:view(-1, 1) end
  local spec = (torch.type(model) == 'bot7.models.gp_hip') and score_spec(self.score, Y_obs, keep) or nil
  if spec == nil then return parent.nominate(self, candidates) end

  local X_obs, S = self.observed, self.config.bot.nSamples
  model:sample_hypers(X_obs, Y_obs)                                          -- :68
  local hyps = ffi.new('b7_hyp[?]', S)
  for s = 1, S do                                                            -- :73-75
    local hyp = model:parse_hypers(model:sample_hypers(X_obs, Y_obs, nil, nil, true))
    local ls  = hip.pin(hyp.lenscale_sq)
    keep[#keep + 1] = ls                                                     -- alive until the call has returned
    hyps[s-1].lenscale_sq, hyps[s-1].amp, hyps[s-1].noise, hyps[s-1].mean = hip.data(ls), hyp.amp, hyp.noise, hyp.mean
  end
  if candidates ~= nil then
    model:stage(X_obs, Y_obs, candidates)                                    -- data + grid resident; uploads what changed
  else
    model:stage_data(X_obs, Y_obs)   -- this rank's shard has run empty: it still refits and takes part in the exchange
  end
  local v, i = ffi.new('double[1]'), ffi.new('int64_t[1]')
  local jit, info = ffi.new('double[?]', S), ffi.new('int[?]', S)
  if hip.group then   -- one process, several GPUs: i indexes self.candidates, the host tensor of all candidates
    hip.gcheck(hip.C.b7_group_eval_nominate(hip.group, S, hyps, spec, v, i, jit, info))
  else                -- i is 1-based in the union of the ranks' shards (= in self.candidates when there is one rank)
    hip.check(hip.C.b7_eval_nominate(hip.ctx, S, hyps, spec, D.lo, v, i, jit, info))   -- :76-79 + :96
  end
  for s = 0, S - 1 do
    if jit[s] > 0 then   -- the reference's warning text, utils/math.lua:210-212
      print(string.format('Warning: utils.math.chol succeeded in factorizing the\ninput matrix after applying a jitter of %.2e', jit[s]))
    end
  end
  self.best_score = v[0]
  return torch.LongTensor{tonumber(i[0])}
end

-- bots/abstract.lua:112-152 for candidates sharded one process per GPU.  Only line 118 differs: `idx` is 1-based in the
-- union of the shards, so it must not index this rank's shard; dist_hip.commit returns the nominee on every rank and
-- deletes the row where it lives.  With one rank (or a group) the parent's code is right as it stands.
function bot:run_trial()
  if D.world == 1 then return parent.run_trial(self) end
  local utils = require('bot7.utils')
  self.nTrials = self.nTrials + 1                                            -- :114
  local idx = self:nominate()                                                -- :117, an index into the UNION
  local g   = torch.isTensor(idx) and idx:view(-1)[1] or idx
  local row, loc = D.commit(g, self.candidates and self.candidates:size(2) or nil)   -- :118 across ranks
  self.pending = utils.tensor.append(self.pending, row:view(1, -1))
  if loc > 0 then   -- this rank held it: the host copy of the shard follows the device (same stable deletion)
    self.candidates = utils.tensor.remove(self.candidates, torch.LongTensor{loc})
    if self.candidates ~= nil then hip.set_resident(self.candidates) else hip.forget_resident() end
  end
  idx = self.pending:size(1)                                                 -- :120
  local nominee = self.pending:select(1, idx)                                -- :121
  local y = self.objective(nominee)                                          -- :124 (every rank: lock step)
  if not torch.isTensor(y) then                                              -- :127-133
    if type(y) == 'number' then y = torch.Tensor{{y}} elseif type(y) == 'table' then y = torch.Tensor(y) end
  end
  if y:dim() == 1 then y:resize(y:nElement(), 1) end                         -- :134
  if self.nTrials == 1 then self.responses = y else self.responses = self.responses:cat(y, 1) end   -- :137-141
  self.observed, self.pending = utils.tensor.steal(self.observed, self.pending, torch.LongTensor{idx})   -- :143-144
  if self.model and self.nTrials == self.config.bot.nInitial then            -- :147-149
    self.model:init(self.observed, self.responses)
  end
  return nominee, y
end

return bot
