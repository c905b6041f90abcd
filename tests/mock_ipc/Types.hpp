// TEST INFRASTRUCTURE -- stand-in for src/Utils/Types.hpp: the reference is compiled with DIM = 3 for every scene this repository covers.
#pragma once
#ifndef DIM
#define DIM 3
#endif
