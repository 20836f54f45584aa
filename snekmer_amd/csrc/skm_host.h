// Error plumbing shared by every translation unit, free of HIP: skm_host.cpp (plain C++, also built with
// -fsanitize=address,undefined by `make -C oracle asan`) and, through skm_common.h, the .hip files.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "snekmer_hip.h"

void skm_set_error(const char *fmt, ...);

#define SKM_REQUIRE(cond, code, ...)   \
    do {                               \
        if (!(cond)) {                 \
            skm_set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

#define SKM_TRY(expr)          \
    do {                       \
        int _s = (expr);       \
        if (_s != SKM_OK)      \
            return _s;         \
    } while (0)
