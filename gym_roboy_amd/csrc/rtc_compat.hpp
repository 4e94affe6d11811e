// rtc_compat.hpp - the few standard declarations the device headers use, for both compilers: hipcc (the
// library build: the real headers) and hiprtc (run-time specialisation, msj_jit.hpp: no standard library
// headers are reachable there, so the fixed-width integer names and std::is_same are declared by hand).
#pragma once
#if defined(__HIPCC_RTC__)
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef __UINTPTR_TYPE__ uintptr_t;
namespace std {
template <class A, class B> struct is_same { static constexpr bool value = false; };
template <class A> struct is_same<A, A> { static constexpr bool value = true; };
}  // namespace std
#else
#include <cmath>
#include <cstdint>
#include <type_traits>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#endif
