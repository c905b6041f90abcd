// TEST INFRASTRUCTURE -- pre-included (g++ -include) in front of the reference's own src/main.cpp, compiled where it lies.
// It stands for the ONE line a maintainer changes in the reference, src/main.cpp:1397:
//     optimizer = new IPC::Optimizer<DIM>(...)   ->   optimizer = new IPC::HipOptimizer<DIM>(...)
// (and the declaration `IPC::Optimizer<DIM>* optimizer;` at :38 keeps compiling either way).  Every header main.cpp includes is
// pulled in first, so their include guards make main.cpp's own includes no-ops; only then is the name redirected.
#pragma once
#include "Types.hpp"
#include "IglUtils.hpp"
#include "Config.hpp"
#include "Optimizer.hpp"
#include "NeoHookeanEnergy.hpp"
#include "FixedCoRotEnergy.hpp"
#include "GIF.hpp"
#include "Timer.hpp"
#include "getRSS.hpp"
#include "CCDUtils.hpp"
#include <igl/readOBJ.h>
#include <igl/colormap.h>
#include <sys/stat.h>
#include <fstream>
#include <string>
#include <ctime>
#include <ghc/fs_std.hpp>
#include <spdlog/spdlog.h>
#include <CLI/CLI.hpp>
#include <tbb/info.h>
#include <tbb/global_control.h>

#include "HipOptimizer.hpp"
#define Optimizer HipOptimizer
