--[[
Drop-in for bot7.grids.sobol (grids/sobol.lua) backed by b7_grid_sobol.
Register:  bot7.grids.sobol = require('bot7hip.grids_sobol_hip')   (or select with config.grid.type = 'sobol_hip')
Same constructor asserts (grids/sobol.lua:35-36) and the same generate(config) contract (:58-90); the grid also
stays resident on the GPU as the candidate set.
--]]
local ffi = require('ffi')
local hip = require('bot7hip.bot7hip_ffi')

local title  = 'bot7.grids.sobol_hip'
local parent = 'bot7.grids.abstract'
local grid, parent = torch.class(title, parent)

function grid:__init(config)
  parent.__init(self)
  local C = config or {}
  C.max_dims = C.max_dims or 40
  C.log_max  = C.log_max or 30
  assert(C.size)
  assert(C.dims and C.dims < C.max_dims)
  self.config = C
end

function grid:generate(config)
  local config = config or self.config
  local skip   = config.skip or 1                 -- :70 (0 is truthy in Lua: skip = 0 stays 0)
  local out    = torch.DoubleTensor(config.size, config.dims)
  -- both, one or none of mins / maxes: the library applies grids/sobol.lua:79-85 as written (the one-sided maps use the
  -- column extremes of the WHOLE grid, combined across ranks / group members inside the call)
  local mins, maxes = hip.pin(config.mins), hip.pin(config.maxes)
  if hip.group then
    hip.gcheck(hip.C.b7_group_grid_sobol(hip.group, config.size, config.dims, skip, hip.data(mins), hip.data(maxes)))
    hip.gcheck(hip.C.b7_group_grid_download(hip.group, 0, config.size, torch.data(out)))
  else
    hip.check(hip.C.b7_grid_sobol(hip.ctx, config.size, config.dims, skip, hip.data(mins), hip.data(maxes), torch.data(out)))
  end
  hip.set_resident(out)
  return out
end

return grid
