// ORACLE / TEST INFRASTRUCTURE: see parallel_for.h
#pragma once
#include "parallel_for.h"
#include "info.h"
