// TEST INFRASTRUCTURE -- pre-included (g++ -include) in front of the reference's own src/TimeStepper/Optimizer.cpp, compiled where it lies.
// It stands for the two includes a maintainer adds to Optimizer.cpp behind its other includes: `#include "HipSelfCollisionHandler.hpp"` and
// `#include "HipSelfCollisionHandlerRedirect.hpp"`.  The
// guarded headers of Optimizer.cpp that mention the handler are pulled in first (their include guards make Optimizer.cpp's own includes no-ops);
// only then is the name `SelfCollisionHandler` redirected (HipSelfCollisionHandlerRedirect.hpp), so that the 44 call sites of the unchanged
// Optimizer.cpp reach the device-side statics.
#pragma once
#include "Optimizer.hpp"
#include "SelfCollisionHandler.hpp"
#include "LinSysSolver.hpp"
#include "HipSelfCollisionHandler.hpp"
#include "HipSelfCollisionHandlerRedirect.hpp"
