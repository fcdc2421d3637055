--[[
Drop-in for bot7.grids.random (grids/random.lua:23-35) backed by b7_grid_random.
Register:  bot7.grids.random_hip = require('bot7hip.grids_random_hip')   (config.grid.type = 'random_hip')
Two streams (config.stream):
  'torch' (default)  the uniforms are torch.rand(size, dims), Torch's own generator in its current state -- line 24 of the
                     reference as it stands, so a seeded reference run is reproduced point for point; the map of :27-33 runs
                     on the host tensor as the reference writes it and the result is uploaded (once per grid).
  'counter'          the library's counter-based generator (config.seed, default 0), made on the GPU; a rank that owns rows
                     [lo, hi) of a global grid passes config.row_offset = lo (a sequential stream cannot be sharded), and
                     the one-sided maps use the column extremes of the whole grid across ranks.
The grid stays resident on the GPU either way.  (Outside Lua the same Torch stream is b7_grid_random_torch.)
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
  if (config.stream or 'torch') == 'torch' and not config.row_offset then
    local out = torch.rand(config.size, config.dims)                           -- :24
    if config.mins and config.maxes then                                       -- :26-28
      out:cmul(out, torch.add(config.maxes, -config.mins):expandAs(out)):add(config.mins:expandAs(out))
    elseif config.mins then                                                    -- :29-30
      out:add(torch.add(config.mins, out:min(1)[1]):expandAs(out))
    elseif config.maxes then                                                   -- :31-32
      out:cmul(torch.cdiv(config.maxes, out:max(1)[1]):expandAs(out))
    end
    hip.upload_grid(out)                                                       -- to the GPU (sharded over a group's members)
    return out
  end
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
