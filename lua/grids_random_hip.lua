--[[
Drop-in for bot7.grids.random (grids/random.lua:23-35) backed by b7_grid_random.
Register:  bot7.grids.random_hip = require('bot7hip.grids_random_hip')   (config.grid.type = 'random_hip')
torch.rand's MT19937 stream is not part of the reference tree, so the uniforms come from the library's counter-based
generator (config.seed, default 0; a rank that owns rows [lo, hi) of a global grid passes config.row_offset = lo);
the affine map and its one-sided branches are the reference's (:27-33).  The grid stays resident on the GPU.
--]]
local ffi = require('ffi')
local hip = require('bot7hip.bot7hip_ffi')

local title  = 'bot7.grids.random_hip'
local parent = 'bot7.grids.abstract'
local grid, parent = torch.class(title, parent)

function grid:__init(config)
  parent.__init(self)
  self.config = config or {}
end

function grid:generate(config)
  local config = config or self.config
  local out    = torch.DoubleTensor(config.size, config.dims)
  local mins, maxes = hip.pin(config.mins), hip.pin(config.maxes)            -- :27-33, one-sided maps included
  if hip.group then
    hip.gcheck(hip.C.b7_group_grid_random(hip.group, config.size, config.dims, config.seed or 0, hip.data(mins), hip.data(maxes)))
    hip.gcheck(hip.C.b7_group_grid_download(hip.group, 0, config.size, torch.data(out)))
  else
    hip.check(hip.C.b7_grid_random(hip.ctx, config.size, config.dims, config.seed or 0, config.row_offset or 0,
                                   hip.data(mins), hip.data(maxes), torch.data(out)))            -- :24
  end
  hip.set_resident(out)
  return out
end

return grid
