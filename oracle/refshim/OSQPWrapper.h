// ORACLE / TEST INFRASTRUCTURE: the reference's Optimizer owns an OSQP wrapper object for its SQP / QP baselines
// (out of scope).  This stand-in has the members Optimizer.cpp mentions; using it aborts.
#pragma once
#include <cstdlib>
#include <iostream>
typedef int c_int;
typedef double c_float;
class OSQP {
public:
    OSQP(bool = false) {}
    void setup(c_float*, c_int, c_int*, c_int*, c_float*, c_float*, c_int, c_int*, c_int*, c_float*, c_float*, c_int, c_int)
    {
        std::cerr << "refshim: OSQP is not provided (QP / SQP baselines are out of scope)" << std::endl;
        std::abort();
    }
    c_int solve() { std::abort(); return 0; }
    c_float* getPrimal() const { return nullptr; }
    c_float* getDual() const { return nullptr; }
};
