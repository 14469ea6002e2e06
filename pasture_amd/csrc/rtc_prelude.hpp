// What the device headers need from <stdint.h>, <type_traits> and <limits> when they are compiled by hipRTC at run time (jit.cpp):
// hipRTC brings the HIP device runtime with it but no C or C++ standard library headers.  Included only under __HIPCC_RTC__
// (device_common.hpp); the in-tree build uses the real headers.
#pragma once
#ifdef __HIPCC_RTC__

typedef unsigned char uint8_t;
typedef signed char int8_t;
typedef unsigned short uint16_t;
typedef short int16_t;
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned long long uint64_t;
typedef long long int64_t;
typedef unsigned long uintptr_t;

namespace std {

template <typename T, T V> struct integral_constant { static constexpr T value = V; };
typedef integral_constant<bool, true> true_type;
typedef integral_constant<bool, false> false_type;

template <typename A, typename B> struct is_same : false_type {};
template <typename A> struct is_same<A, A> : true_type {};

template <bool C, typename A, typename B> struct conditional { typedef A type; };
template <typename A, typename B> struct conditional<false, A, B> { typedef B type; };

template <typename T> struct is_floating_point : false_type {};
template <> struct is_floating_point<float> : true_type {};
template <> struct is_floating_point<double> : true_type {};

template <typename T> struct is_integral : false_type {};
template <> struct is_integral<uint8_t> : true_type {};
template <> struct is_integral<int8_t> : true_type {};
template <> struct is_integral<uint16_t> : true_type {};
template <> struct is_integral<int16_t> : true_type {};
template <> struct is_integral<uint32_t> : true_type {};
template <> struct is_integral<int32_t> : true_type {};
template <> struct is_integral<uint64_t> : true_type {};
template <> struct is_integral<int64_t> : true_type {};

template <typename T> struct is_signed : integral_constant<bool, (T(-1) < T(0))> {};
template <typename T> struct is_unsigned : integral_constant<bool, is_integral<T>::value && !(T(-1) < T(0))> {};

template <typename T> struct make_unsigned { typedef T type; };
template <> struct make_unsigned<int8_t> { typedef uint8_t type; };
template <> struct make_unsigned<int16_t> { typedef uint16_t type; };
template <> struct make_unsigned<int32_t> { typedef uint32_t type; };
template <> struct make_unsigned<int64_t> { typedef uint64_t type; };

// integers only (the one use: saturating float -> int of Rust `as`)
template <typename T> struct numeric_limits {
  static constexpr bool sgn = (T(-1) < T(0));
  __host__ __device__ static constexpr T max() { return sgn ? (T)(((uint64_t)1 << (sizeof(T) * 8 - 1)) - 1) : (T)~(T)0; }
  __host__ __device__ static constexpr T min() { return sgn ? (T)((uint64_t)1 << (sizeof(T) * 8 - 1)) : (T)0; }
};

}  // namespace std

#endif  // __HIPCC_RTC__
