// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wavefront = 64 lanes everywhere in this directory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvl {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;   // one 16-byte chunk
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;   // one 8-byte chunk

constexpr int kWave = 64;

// ---------------------------------------------------------------- dtype traits
// CODE 0 = bfloat16, 1 = float16.  Storage is handled as raw 16-bit words.
template <int CODE> struct DType;

template <> struct DType<0> {
    using vec8 = bf16x8_t;
    static __device__ __forceinline__ float to_float(uint16_t v) {
        return __uint_as_float(((uint32_t)v) << 16);
    }
    static __device__ __forceinline__ uint16_t from_float(float f) {
        __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32, NaN-preserving
        return __builtin_bit_cast(uint16_t, b);
    }
    static __device__ __forceinline__ f32x4_t mfma16(vec8 a, vec8 b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16_t mfma32(vec8 a, vec8 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

template <> struct DType<1> {
    using vec8 = f16x8_t;
    static __device__ __forceinline__ float to_float(uint16_t v) {
        return (float)__builtin_bit_cast(_Float16, v);
    }
    static __device__ __forceinline__ uint16_t from_float(float f) {
        _Float16 h = (_Float16)f;
        return __builtin_bit_cast(uint16_t, h);
    }
    static __device__ __forceinline__ f32x4_t mfma16(vec8 a, vec8 b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16_t mfma32(vec8 a, vec8 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// two floats -> one packed 32-bit word (element 0 in the low half): a single
// v_cvt_pk_bf16_f32 / v_cvt_pkrtz-free f16 pack instead of two converts and an OR
template <int CODE>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t f = {lo, hi};
    if (CODE == 0) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
    } else {
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2_t));
    }
}

// c + sum of the squares of the two 16-bit values packed in w (one v_dot2c instruction)
template <int CODE>
__device__ __forceinline__ float sumsq2(uint32_t w, float c) {
    if (CODE == 0) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
        const bf16x2_t v = __builtin_bit_cast(bf16x2_t, w);
        return __builtin_amdgcn_fdot2_f32_bf16(v, v, c, false);
    } else {
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
        const f16x2_t v = __builtin_bit_cast(f16x2_t, w);
        return __builtin_amdgcn_fdot2(v, v, c, false);
    }
}

template <int CODE>
__device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
    lo = DType<CODE>::to_float((uint16_t)(w & 0xffffu));
    hi = DType<CODE>::to_float((uint16_t)(w >> 16));
}

template <int CODE>
__device__ __forceinline__ void unpack8(u32x4_t v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2<CODE>(v[i], f[2 * i], f[2 * i + 1]);
}

template <int CODE>
__device__ __forceinline__ u32x4_t pack8(const float (&f)[8]) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2<CODE>(f[2 * i], f[2 * i + 1]);
    return v;
}

// -------------------------------------------------------------- wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum over all threads of a block (<= 1024 threads).  `scratch` holds >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    if (nw == 1) return v;
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float t = (lane < nw) ? scratch[lane] : 0.f;
    t = wave_sum(t);
    __syncthreads();
    return t;
}

// ------------------------------------------------------------- LDS tile image
// Byte offset of 16-byte chunk `ch` of row `row` in a [rows][CPR chunks] tile whose rows
// are CPR*16 bytes.  The XOR keeps ds_read_b128 row reads (MFMA A/B fragments, lanes =
// rows) and ds_read_b64_tr_b16 transposed reads (4 rows x 16 columns per 16 lanes) off
// each other's banks for 256-byte rows.
template <int CPR>
__device__ __forceinline__ int tile_off(int row, int ch) {
    const int x = (((row & 3) << 2) | ((row >> 2) & 3)) & (CPR - 1);
    return row * (CPR * 16) + ((ch ^ x) << 4);
}

__device__ __forceinline__ s16x4_t lds_read_tr16(const char* lds_base, int byte_off) {
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_ptr_t)(__attribute__((address_space(3))) char*)(lds_base + byte_off));
}

// Asynchronous 16-byte-per-lane global -> LDS copy (LDS-DMA).  `lds_wave_base` must be the same
// for every lane of the wave: lane l's 16 bytes land at lds_wave_base + 16 l.  The source
// address is per lane.
__device__ __forceinline__ void glds16(const void* gptr, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gptr,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// 16 fp8 (e4m3) values -> 16 values of the 16-bit type CODE, as two 16-byte chunks.  The
// scaled conversion with scale 1 is exact (every e4m3 value is a bf16 and an fp16 value) and
// one instruction per pair, half of fp8 -> fp32 -> 16 bits.
template <int CODE>
__device__ __forceinline__ uint32_t widen_fp8x2(uint32_t v, bool high_word) {
    if constexpr (CODE == 0) {
        return high_word
            ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, true))
            : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, false));
    } else {
        return high_word
            ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v, 1.0f, true))
            : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v, 1.0f, false));
    }
}

template <int CODE>
__device__ __forceinline__ void widen_fp8x16(u32x4_t v, u32x4_t& lo, u32x4_t& hi) {
    uint32_t o[8];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        o[2 * w] = widen_fp8x2<CODE>(v[w], false);
        o[2 * w + 1] = widen_fp8x2<CODE>(v[w], true);
    }
    lo = u32x4_t{o[0], o[1], o[2], o[3]};
    hi = u32x4_t{o[4], o[5], o[6], o[7]};
}

constexpr float kLog2e = 1.4426950408889634f;

}  // namespace mvl

#define MVL_CHECK_LAUNCH() ((int)hipGetLastError())
