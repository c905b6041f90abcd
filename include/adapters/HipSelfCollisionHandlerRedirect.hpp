// The one line that makes the reference's Optimizer.cpp call the device-side statics of HipSelfCollisionHandler.hpp: from here on
// `SelfCollisionHandler<dim>::f(...)` names that class (and, for everything it does not define, the reference's class through it).
// Include it behind HipSelfCollisionHandler.hpp and behind every header that declares or specialises the reference's class.
// Deliberately WITHOUT an include guard: a guarded header that had been seen earlier in the translation unit -- HipOptimizer.hpp includes
// the class -- would make a later include, meant to switch the redirect on, a silent no-op and leave all 44 call sites on the host code.
#ifndef IPCGPU_HIP_SELF_COLLISION_HANDLER_DECLARED
#error "include HipSelfCollisionHandler.hpp before HipSelfCollisionHandlerRedirect.hpp"
#endif
#undef SelfCollisionHandler
#define SelfCollisionHandler HipSelfCollisionHandler
