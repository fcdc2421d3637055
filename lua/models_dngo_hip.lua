--[[
Drop-in for bot7.models.dngo's predict path (models/dngo.lua:108-175) backed by b7_blr_fit_x / b7_blr_basis /
b7_blr_predict.  The network is still built and trained by nnTools exactly as in the reference (:49-106,
:126-152): this class only takes the trained nn.Linear layers up to the basis layer out of `self.network`,
hands them to the GPU as plain arrays, and replaces the minibatch forward loops (:155-171) and the
Bayesian-linear head (:174).  Register:  bot7.models.dngo_hip = require('bot7hip.models_dngo_hip')
--]]
local ffi = require('ffi')
local hip = require('bot7hip.bot7hip_ffi')

local title  = 'bot7.models.dngo_hip'
local parent = 'bot7.models.dngo'
local dngo, parent = torch.class(title, parent)

local ACT = {['nn.Tanh'] = 1, ['nn.ReLU'] = 2, ['nn.Sigmoid'] = 3}

-- Collect nn.Linear weights/biases from the input up to self.basis (the module whose output is the feature map)
local function pack_network(self)
  local Ws, bs, act = {}, {}, 0
  for i = 1, self.network:size() do
    local mod = self.network:get(i)
    if torch.type(mod) == 'nn.Linear' then
      Ws[#Ws+1] = mod.weight:double():contiguous(); bs[#bs+1] = mod.bias:double():contiguous()
    elseif ACT[torch.type(mod)] then
      act = ACT[torch.type(mod)]
    end
    if mod == self.basis then break end
  end
  local n    = #Ws
  local dims = ffi.new('int[?]', n + 1)
  local Wp, bp = ffi.new('const double*[?]', n), ffi.new('const double*[?]', n)
  dims[0] = Ws[1]:size(2)
  for l = 1, n do dims[l] = Ws[l]:size(1); Wp[l-1] = torch.data(Ws[l]); bp[l-1] = torch.data(bs[l]) end
  local net = ffi.new('b7_mlp', {n, dims, Wp, bp, act})
  return net, {Ws, bs, dims, Wp, bp}   -- second value keeps the arrays alive
end

function dngo:predict(X0, Y0, X1, hyp, req, skip)
  if not skip then   -- refresh the network on (X0, Y0) with nnTools.trainer, as the reference does before predicting (:126-152)
    local trainer = require('bot7.nnTools.trainer')
    if self.state.dfdx then self.state.dfdx:zero() end
    trainer(self.network, {xr = X0, yr = Y0}, self.config.update,
            {optimizer = self.optimizer, buffers = self.buffers, criterion = self.criterion, state = self.state})
  end
  self.network:evaluate()
  local net, keep = pack_network(self)
  local z  = keep[3][#keep[1]]
  local h  = self.blr_hyp or {alpha = 1.0, beta = 1.0 / (1e-2 * Y0:var()), mean = Y0:mean()}
  -- :155-162 (features of X0) and the fit half of :174 in one call; Z0 never leaves the device
  local X0c, Y0c = hip.pin(X0), hip.pin(Y0)
  hip.check(hip.C.b7_blr_fit_x(hip.ctx, net, hip.data(X0c), hip.data(Y0c), X0:size(1), h.alpha, h.beta, h.mean, nil))
  assert(hip.group == nil, 'bot7hip: the DNGO head runs on one GPU (no group)')
  if not hip.is_resident(X1) then hip.upload_grid(X1) end
  hip.check(hip.C.b7_blr_basis(hip.ctx, net, nil, 0, nil))                                        -- :164-171
  local mean, var = torch.DoubleTensor(X1:size(1), 1), torch.DoubleTensor(X1:size(1))
  hip.check(hip.C.b7_blr_predict(hip.ctx, torch.data(mean), torch.data(var)))                     -- :174
  return {mean = mean, var = var}
end

-- hyp = 'marginalize' (models/dngo.lua:109, handed to the predictor at :174) as ONE library call, b7_blr_eval_nominate_marg: `hyps`
-- is a list of {alpha=, beta=, mean=} tables (drawn by the host, e.g. bot7.samplers.slice over the evidence); S heads over the
-- same features, the acquisition of every head added in sample order, score:div(S), score:max(1).  spec: a b7_score_spec
-- (scores_hip.lua builds it).  Returns the 1-based index of the nominee in X1 (the resident grid) and its score.
function dngo:eval_nominate(X0, Y0, X1, hyps, spec)
  self.network:evaluate()
  local net, keep = pack_network(self)
  local S = #hyps
  local a, b, m = ffi.new('double[?]', S), ffi.new('double[?]', S), ffi.new('double[?]', S)
  for s = 1, S do a[s-1] = hyps[s].alpha; b[s-1] = hyps[s].beta; m[s-1] = hyps[s].mean end
  assert(hip.group == nil, 'bot7hip: the DNGO head runs on one GPU (no group)')
  if not hip.is_resident(X1) then hip.upload_grid(X1) end
  local X0c, Y0c = hip.pin(X0), hip.pin(Y0)
  local val, idx, jit = ffi.new('double[1]'), ffi.new('int64_t[1]'), ffi.new('double[1]')
  hip.check(hip.C.b7_blr_eval_nominate_marg(hip.ctx, net, hip.data(X0c), hip.data(Y0c), X0:size(1), S, a, b, m, spec, 0, val, idx,
                                            nil, jit))
  return tonumber(idx[0]), val[0]
end

return dngo
