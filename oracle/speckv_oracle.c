/*
 * oracle/speckv_oracle.c -- CPU restatement of the FastLM/CXL-SpecKV hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see speckv_oracle.h).  Plain C99 (single thread; the batch drivers at the
 * end run the per-block functions on several),
 * written from the behaviour of the reference (citations are file:line under
 * /root/reference); no reference source is copied.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * The float paths rely on IEEE single arithmetic with one rounding per
 * operation, which is what the reference gets from g++ -O2 on x86-64 (SSE2).
 */
#define _POSIX_C_SOURCE 200809L
#include "speckv_oracle.h"

#define ORC_STACK_ELEMS 4096   /* one KV block is 2048 elements */

#include <fcntl.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ================================================================== */
/* fp16                                                                */
/* ================================================================== */
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float orc_half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp  = (h >> 10) & 0x1Fu;
    uint32_t man  = h & 0x3FFu;
    if (exp == 0) {
        if (man == 0) return u2f(sign);
        /* subnormal: man * 2^-24 */
        float v = (float)man * (1.0f / 16777216.0f);
        return u2f(f2u(v) | sign);
    }
    if (exp == 31) return u2f(sign | 0x7F800000u | (man << 13));
    return u2f(sign | ((exp + 112u) << 23) | (man << 13));
}

uint16_t orc_float_to_half(float f)
{
    uint32_t u = f2u(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) {               /* inf / nan */
        if (a == 0x7F800000u) return (uint16_t)(sign | 0x7C00u);
        uint32_t m = (a >> 13) & 0x3FFu;
        return (uint16_t)(sign | 0x7C00u | m | 0x200u); /* quiet */
    }
    if (a >= 0x477FF000u)                 /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7C00u);
    if (a < 0x33000001u)                  /* <= 2^-25 -> 0 (tie to even) */
        return (uint16_t)sign;
    int32_t e = (int32_t)(a >> 23) - 127;
    uint32_t m = (a & 0x7FFFFFu) | 0x800000u;  /* 24-bit significand */
    uint32_t shift, half;
    if (e < -14) {                        /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        half = 0;
    } else {
        shift = 13;
        half = (uint32_t)(e + 15) << 10;
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    if (e < -14) return (uint16_t)(sign | q);          /* carry into exp is fine */
    /* q has the implicit bit at position 10 */
    return (uint16_t)(sign | (half + q - 0x400u));
}

/* ================================================================== */
/* host allocator ids                                                  */
/* ================================================================== */
uint64_t orc_virt_page_id(uint64_t handle, uint64_t i)
{   /* speckv_allocator.cpp:24 */
    return (handle << 32) | (i << 12);
}
uint64_t orc_phys_page_id(uint64_t handle, uint64_t i)
{   /* speckv_allocator.cpp:25 */
    return 0x4000000000ULL + (handle << 20) + (i << 12);
}
uint64_t orc_num_pages(uint64_t bytes)
{   /* speckv_allocator.cpp:18-19 */
    return (bytes + ORC_PAGE_SIZE - 1) / ORC_PAGE_SIZE;
}
uint64_t orc_desc_gpu_addr(uint64_t virt_page_id)
{   /* speckv_allocator.cpp:123 */
    return 0x8000000000ULL + (virt_page_id & 0xFFFFFFFFFFFFULL);
}
uint64_t orc_encode_virt_page(uint32_t req_id, uint16_t layer, uint16_t head,
                              uint32_t pos, uint8_t kind)
{   /* speckv_allocator.cpp:92-103 */
    return ((uint64_t)req_id << 32) | ((uint64_t)layer << 16) |
           ((uint64_t)head << 8) | ((uint64_t)pos << 1) | (uint64_t)kind;
}

/* ---- C ABI model -------------------------------------------------- */
typedef struct {
    uint64_t handle;
    uint64_t size_bytes;
    uint64_t n_pages;
    uint32_t* flags;     /* page_table_ copy flags (speckv_allocator.cpp:135) */
    int live;
} orc_allocation_t;

struct orc_cabi {
    int initialized;
    uint64_t next_handle;
    orc_allocation_t* allocs;
    size_t n_allocs, cap_allocs;
};

orc_cabi_t* orc_cabi_new(void)
{
    orc_cabi_t* c = (orc_cabi_t*)calloc(1, sizeof(*c));
    c->next_handle = 1;
    return c;
}
static void cabi_drop_all(orc_cabi_t* c)
{
    for (size_t i = 0; i < c->n_allocs; ++i) free(c->allocs[i].flags);
    free(c->allocs);
    c->allocs = NULL; c->n_allocs = c->cap_allocs = 0;
    c->next_handle = 1;   /* a new SpeckvAllocator starts at 1 (speckv_allocator.hpp:57) */
}
void orc_cabi_delete(orc_cabi_t* c) { if (c) { cabi_drop_all(c); free(c); } }

int orc_cabi_init(orc_cabi_t* c, const char* dev_path)
{   /* speckv_c_api.cpp:13-32, speckv_driver.cpp:11-16 */
    if (c->initialized) return -1;
    int fd = dev_path ? open(dev_path, O_RDWR) : -1;
    if (fd < 0) return -1;            /* ctor throws -> catch(...) -> GENERAL */
    close(fd);
    c->initialized = 1;
    return 0;
}
void orc_cabi_finalize(orc_cabi_t* c)
{   /* speckv_c_api.cpp:34-39 */
    cabi_drop_all(c);
    c->initialized = 0;
}
static orc_allocation_t* cabi_find(orc_cabi_t* c, uint64_t h)
{
    for (size_t i = 0; i < c->n_allocs; ++i)
        if (c->allocs[i].live && c->allocs[i].handle == h) return &c->allocs[i];
    return NULL;
}
int orc_cabi_alloc(orc_cabi_t* c, uint64_t bytes, uint64_t* out)
{   /* speckv_c_api.cpp:41-53, speckv_allocator.cpp:11-38 */
    if (!c->initialized || !out) return -4;
    if (c->n_allocs == c->cap_allocs) {
        c->cap_allocs = c->cap_allocs ? 2 * c->cap_allocs : 16;
        c->allocs = (orc_allocation_t*)realloc(c->allocs, c->cap_allocs * sizeof(orc_allocation_t));
    }
    orc_allocation_t* a = &c->allocs[c->n_allocs++];
    a->handle = c->next_handle++;
    a->size_bytes = bytes;
    a->n_pages = orc_num_pages(bytes);
    a->flags = (uint32_t*)calloc(a->n_pages ? a->n_pages : 1, sizeof(uint32_t));
    a->live = 1;
    *out = a->handle;
    return 0;
}
int orc_cabi_free(orc_cabi_t* c, uint64_t h)
{   /* speckv_c_api.cpp:55-64, speckv_allocator.cpp:40-52 : unknown -> OK */
    if (!c->initialized) return -4;
    orc_allocation_t* a = cabi_find(c, h);
    if (a) { a->live = 0; free(a->flags); a->flags = NULL; }
    return 0;
}
int orc_cabi_access(orc_cabi_t* c, uint64_t h, uint64_t off, uint64_t len, uint64_t* out)
{   /* speckv_c_api.cpp:66-83, speckv_allocator.cpp:54-74 ; len ignored */
    (void)len;
    if (!c->initialized || !out) return -4;
    orc_allocation_t* a = cabi_find(c, h);
    if (!a) return -1;
    uint64_t page_idx = off / ORC_PAGE_SIZE, page_off = off % ORC_PAGE_SIZE;
    if (page_idx >= a->n_pages) return -1;
    if ((a->flags[page_idx] & 0x3u) == 0) a->flags[page_idx] |= 0x2u; /* sync fetch -> L2 */
    *out = orc_phys_page_id(h, page_idx) + page_off;
    return 0;
}
int orc_cabi_prefetch(orc_cabi_t* c, uint32_t req_id, uint16_t layer, uint32_t cur_pos,
                      uint32_t depth_k, const int32_t* tokens, uint32_t history_len)
{   /* speckv_c_api.cpp:85-99 ; driver result ignored (speckv_allocator.cpp:89) */
    (void)req_id; (void)layer; (void)cur_pos; (void)depth_k;
    if (!c->initialized || !tokens || history_len == 0) return -4;
    return 0;
}
int orc_cabi_set_prefetch_depth(orc_cabi_t* c, uint32_t k)
{   /* speckv_c_api.cpp:101-110 ; ioctl on the fake device fails -> DRIVER */
    (void)k;
    if (!c->initialized) return -4;
    return -2;
}
int orc_cabi_set_compression_scheme(orc_cabi_t* c, int scheme)
{   /* speckv_c_api.cpp:112-121 */
    (void)scheme;
    if (!c->initialized) return -4;
    return -2;
}
int orc_cabi_page_flags(orc_cabi_t* c, uint64_t h, uint64_t page_idx, uint32_t* out)
{
    orc_allocation_t* a = cabi_find(c, h);
    if (!a || page_idx >= a->n_pages) return -1;
    *out = a->flags[page_idx];
    return 0;
}

uint64_t orc_calc_offset(uint64_t req_id, uint64_t layer, uint64_t head, uint64_t pos,
                         uint64_t kind, uint64_t entry_bytes, uint64_t num_layers,
                         uint64_t num_tokens, uint64_t num_heads)
{   /* vllm_speckv_backend.py:95-100 */
    return ((((req_id * num_layers + layer) * 2 + kind) * num_tokens + pos) * num_heads + head)
           * entry_bytes;
}
uint64_t orc_shim_total_bytes(uint64_t T, uint64_t L, uint64_t H, uint64_t D, uint64_t bpe)
{   /* vllm_speckv_backend.py:39-40 */
    return T * L * H * D * bpe * 2;
}

/* ================================================================== */
/* codec                                                               */
/* ================================================================== */
float orc_compute_scale(const float* x, size_t n)
{   /* cache_engine.cpp:172-184 : NaN never wins the '>' compare */
    float max_val = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float a = fabsf(x[i]);
        if (a > max_val) max_val = a;
    }
    return (max_val > 0.0f) ? (max_val / 127.0f) : 1.0f;
}

/* float -> int32 as x86-64 cvttss2si does it: out of range / NaN give
 * INT32_MIN ("integer indefinite").  The reference's static_cast<int8_t>
 * of a float compiles to that instruction followed by a byte truncation
 * (cache_engine.cpp:191). */
static int32_t cvttss2si(float v)
{
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT32_MIN;
    return (int32_t)v;
}

void orc_quantize(const float* x, size_t n, float scale, int mode, int8_t* q)
{
    if (mode == ORC_QUANT_REF_EXACT) {
        /* cache_engine.cpp:189-192 : two rounded ops, roundf (half away),
         * int8 wrap; the following clamp is a no-op on an int8 value. */
        for (size_t i = 0; i < n; ++i) {
            float scaled = x[i] / scale;
            float r = roundf(scaled * 127.0f);
            q[i] = (int8_t)(uint8_t)((uint32_t)cvttss2si(r) & 0xFFu);
        }
    } else {
        /* intent: q = clamp(round(x / s), -127, 127)
         * (cache_engine.cpp:182 comment, kv_compress.v:130) */
        for (size_t i = 0; i < n; ++i) {
            float r = roundf(x[i] / scale);
            if (!(r == r)) r = 0.0f;
            if (r > 127.0f) r = 127.0f;
            if (r < -127.0f) r = -127.0f;
            q[i] = (int8_t)(int32_t)r;
        }
    }
}

void orc_delta_encode(const int8_t* q, size_t n, int8_t* d)
{   /* cache_engine.cpp:198-211 */
    if (n == 0) return;
    d[0] = q[0];
    for (size_t i = 1; i < n; ++i)
        d[i] = (int8_t)(uint8_t)((uint8_t)q[i] - (uint8_t)q[i - 1]);
}

size_t orc_rle_encode(const int8_t* d, size_t n, uint8_t* out)
{   /* cache_engine.cpp:213-239 */
    if (n == 0) return 0;
    size_t w = 0;
    int8_t cur = d[0];
    size_t count = 1;
    for (size_t i = 1; i < n; ++i) {
        if (d[i] == cur && count < 255) {
            count++;
        } else {
            out[w++] = (uint8_t)cur;
            out[w++] = (uint8_t)count;
            cur = d[i];
            count = 1;
        }
    }
    out[w++] = (uint8_t)cur;
    out[w++] = (uint8_t)count;
    return w;
}

size_t orc_rle_decode(const uint8_t* rle, size_t len, int8_t* out, size_t cap, size_t* total)
{   /* cache_engine.cpp:241-258 : odd trailing byte dropped, count 0 emits nothing */
    size_t w = 0, tot = 0;
    for (size_t i = 0; i + 1 < len; i += 2) {
        int8_t v = (int8_t)rle[i];
        uint8_t c = rle[i + 1];
        tot += c;
        for (size_t j = 0; j < c; ++j)
            if (w < cap) out[w++] = v;
    }
    if (total) *total = tot;
    return w;
}

void orc_delta_decode(const int8_t* d, size_t n, int8_t* q)
{   /* cache_engine.cpp:260-273 */
    if (n == 0) return;
    q[0] = d[0];
    for (size_t i = 1; i < n; ++i)
        q[i] = (int8_t)(uint8_t)((uint8_t)q[i - 1] + (uint8_t)d[i]);
}

void orc_dequantize(const int8_t* q, size_t n, float scale, int mode, float* y)
{
    if (mode == ORC_QUANT_REF_EXACT) {
        /* cache_engine.cpp:278-281 : divide by 127 first, then scale */
        for (size_t i = 0; i < n; ++i) {
            float s = (float)q[i] / 127.0f;
            y[i] = s * scale;
        }
    } else {
        for (size_t i = 0; i < n; ++i) y[i] = (float)q[i] * scale;
    }
}

size_t orc_compress_f32(const float* x, size_t n, int mode, float* scale, uint8_t* rle)
{   /* cache_engine.cpp:40-82 */
    float s = orc_compute_scale(x, n);
    *scale = s;
    if (n == 0) return 0;
    int8_t* q = (int8_t*)malloc(n);
    int8_t* d = (int8_t*)malloc(n);
    orc_quantize(x, n, s, mode, q);
    orc_delta_encode(q, n, d);
    size_t len = orc_rle_encode(d, n, rle);
    free(q); free(d);
    return len;
}

size_t orc_decompress_f32(const uint8_t* rle, size_t len, float scale, int mode,
                          float* y, size_t cap)
{   /* cache_engine.cpp:84-116 */
    /* block-sized outputs use the stack: a malloc/free pair per block serialises the threaded batch driver */
    int8_t sd[ORC_STACK_ELEMS], sq[ORC_STACK_ELEMS];
    int8_t* d = cap <= ORC_STACK_ELEMS ? sd : (int8_t*)malloc(cap);
    int8_t* q = cap <= ORC_STACK_ELEMS ? sq : (int8_t*)malloc(cap);
    size_t n = orc_rle_decode(rle, len, d, cap, NULL);
    orc_delta_decode(d, n, q);
    orc_dequantize(q, n, scale, mode, y);
    if (d != sd) free(d);
    if (q != sq) free(q);
    return n;
}

/* ---- e4m3fn (OCP FP8: 1-4-3, bias 7, no inf, 0x7F/0xFF NaN, max 448) ---- */
float orc_e4m3_to_f32(uint8_t b)
{
    uint32_t sign = (b & 0x80u) ? 0x80000000u : 0u;
    uint32_t e = (b >> 3) & 0xFu, m = b & 7u;
    if (e == 15 && m == 7) return u2f(sign | 0x7FC00000u);
    if (e == 0) { float v = (float)m * (1.0f / 512.0f); return u2f(f2u(v) | sign); }   /* m/8 * 2^-6 */
    return u2f(sign | ((e + 120u) << 23) | (m << 20));
}
uint8_t orc_f32_to_e4m3(float f)
{
    uint32_t u = f2u(f);
    uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    uint32_t a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (uint8_t)(sign | 0x7Fu);                 /* NaN */
    if (a >= 0x43E00000u) return (uint8_t)(sign | 0x7Eu);                /* >= 448: saturate */
    if (a < 0x3A800000u) {                                               /* < 2^-10: rounds to 0 or min subnormal */
        /* min subnormal 2^-9; ties at 2^-10 go to even (0) */
        return (uint8_t)(sign | ((a > 0x3A800000u) ? 1u : 0u));
    }
    int32_t e = (int32_t)(a >> 23) - 127;
    uint32_t m = (a & 0x7FFFFFu) | 0x800000u;
    uint32_t shift;
    uint32_t base;
    if (e < -6) { shift = (uint32_t)(20 + (-6 - e)); base = 0; }           /* subnormal target */
    else        { shift = 20; base = (uint32_t)(e + 7) << 3; }
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    uint32_t r = (e < -6) ? q : base + q - 8u;
    if (r > 0x7Eu) r = 0x7Eu;
    return (uint8_t)(sign | r);
}

/* ---- MXFP4 (OCP MX v1.0): E2M1 elements, E8M0 block scales ---------------------------------------------------- */
float orc_e2m1_to_f32(uint8_t nibble)
{
    static const float mag[8] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};
    float v = mag[nibble & 7u];
    return (nibble & 8u) ? -v : v;
}
uint8_t orc_f32_to_e2m1(float v)
{
    /* representable magnitudes 0, .5, 1, 1.5, 2, 3, 4, 6; a value half way between two goes to the one with the even code
     * (0.25 -> 0, 0.75 -> 1, 1.25 -> 1, 1.75 -> 2, 2.5 -> 2, 3.5 -> 4, 5 -> 4); beyond 6: 6 */
    static const float mag[8] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};
    if (v != v) return 0;
    uint8_t sign = (f2u(v) & 0x80000000u) ? 8u : 0u;
    float a = fabsf(v);
    if (a >= 6.0f) return (uint8_t)(sign | 7u);
    uint8_t c = 0;
    for (uint8_t i = 0; i < 7; ++i) {
        if (a >= mag[i] && a <= mag[i + 1]) {
            float mid = 0.5f * (mag[i] + mag[i + 1]);                 /* exact */
            if (a < mid) c = i; else if (a > mid) c = (uint8_t)(i + 1); else c = (i & 1u) ? (uint8_t)(i + 1) : i;
            break;
        }
    }
    return (uint8_t)(sign | c);
}
uint8_t orc_mx_scale_code(float amax, int emax_elem)
{
    if (!(amax > 0.0f)) return 0;
    if (amax > 65504.0f) amax = 65504.0f;
    int ex;
    (void)frexpf(amax, &ex);                                          /* amax = m * 2^ex, m in [0.5, 1): floor(log2) = ex - 1 */
    int code = ex - 1 - emax_elem + 127;
    if (code < 0) code = 0;
    if (code > 254) code = 254;
    return (uint8_t)code;
}
float orc_e8m0_to_f32(uint8_t code)
{
    if (code == 255) return u2f(0x7FC00000u);
    return ldexpf(1.0f, (int)code - 127);
}
static float finite_amax(const float* x, size_t n)
{
    float mx = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float a = fabsf(x[i]);
        if (a != a) continue;                                          /* NaN elements are skipped */
        if (a > 65504.0f) a = 65504.0f;
        if (a > mx) mx = a;
    }
    return mx;
}
/* The two halves of a block are interleaved element by element before the MX conversion: nibble 2i of the stream is element i,
 * nibble 2i+1 element n/2 + i, so byte i = (element i low, element n/2 + i high) and one MX block of 32 consecutive nibbles =
 * 16 elements of the first half + the 16 matching elements of the second (for a page of two positions x 1024 channels: 16
 * channels of both positions -- what one lane of the block-scaled matrix instruction takes as its 32 k values). */
static size_t mx4_elem_of_nibble(size_t nib, size_t n) { return (nib & 1) ? n / 2 + (nib >> 1) : (nib >> 1); }
static size_t compress_mxfp4(const float* xf, size_t n, uint8_t* rec)
{
    size_t groups = n / 32;
    uint8_t* codes = rec + n / 2;
    memset(rec, 0, n / 2 + groups);
    for (size_t g = 0; g < groups; ++g) {
        float blk[32];
        for (size_t i = 0; i < 32; ++i) blk[i] = xf[mx4_elem_of_nibble(g * 32 + i, n)];
        uint8_t code = orc_mx_scale_code(finite_amax(blk, 32), 2);
        codes[g] = code;
        for (size_t i = 0; i < 32; ++i) {
            size_t nib = g * 32 + i;
            float x = blk[i];
            uint8_t q = 0;
            if (x == x) {
                if (x > 65504.0f) x = 65504.0f;
                if (x < -65504.0f) x = -65504.0f;
                q = orc_f32_to_e2m1(ldexpf(x, 127 - (int)code));       /* exact scaling by a power of two */
            }
            rec[nib >> 1] |= (uint8_t)((nib & 1) ? (q << 4) : q);
        }
    }
    return n / 2 + groups;
}
void orc_quantize_rows_mxfp8(const uint16_t* q16, size_t rows, size_t d, size_t block, uint8_t* q8, uint8_t* q_codes)
{
    size_t blocks = d / block;
    for (size_t r = 0; r < rows; ++r)
        for (size_t b = 0; b < blocks; ++b) {
            float x[64];
            for (size_t i = 0; i < block; ++i) x[i] = orc_half_to_float(q16[r * d + b * block + i]);
            uint8_t code = orc_mx_scale_code(finite_amax(x, block), 8);
            q_codes[r * blocks + b] = code;
            for (size_t i = 0; i < block; ++i) {
                float v = x[i];
                uint8_t c = 0;
                if (v == v) {
                    if (v > 65504.0f) v = 65504.0f;
                    if (v < -65504.0f) v = -65504.0f;
                    c = orc_f32_to_e4m3(ldexpf(v, 127 - (int)code));   /* nearest even, saturating at 448 */
                }
                q8[r * d + b * block + i] = c;
            }
        }
}
/* value of channel i of position (2 * page + half) in a page row of d nibble BYTES (byte i = position 2*page low, 2*page+1 high) with d/16 codes */
static double mx4_row_value(const uint8_t* rows, const uint8_t* codes, size_t d, size_t t, size_t i)
{
    size_t page = t >> 1, half = t & 1;
    uint8_t nb = (uint8_t)((rows[page * d + i] >> (half * 4)) & 0xF);
    return (double)orc_e2m1_to_f32(nb) * (double)orc_e8m0_to_f32(codes[page * (d / 16) + i / 16]);
}
void orc_attend_mx4(const uint8_t* q8, const uint8_t* q_codes, size_t q_block, size_t g, const uint8_t* k_rows, const uint8_t* k_codes,
                    const uint8_t* v_rows, const uint8_t* v_codes, size_t n_pos, size_t d, float sm_scale,
                    float* out, float* lse, float* mag)
{
    size_t qblocks = d / q_block;
    double* s = (double*)malloc((n_pos ? n_pos : 1) * sizeof(double));
    double* o = (double*)malloc(d * sizeof(double));
    double* a = (double*)malloc(d * sizeof(double));
    double* qd = (double*)malloc(d * sizeof(double));
    for (size_t m = 0; m < g; ++m) {
        for (size_t i = 0; i < d; ++i)
            qd[i] = (double)orc_e4m3_to_f32(q8[m * d + i]) * (double)orc_e8m0_to_f32(q_codes[m * qblocks + i / q_block]);
        double mx = -INFINITY;
        for (size_t t = 0; t < n_pos; ++t) {
            double acc = 0.0;
            for (size_t i = 0; i < d; ++i) acc += qd[i] * mx4_row_value(k_rows, k_codes, d, t, i);
            s[t] = acc * (double)sm_scale;
            if (s[t] > mx) mx = s[t];
        }
        double l = 0.0;
        for (size_t i = 0; i < d; ++i) { o[i] = 0.0; a[i] = 0.0; }
        for (size_t t = 0; t < n_pos; ++t) {
            double p = exp(s[t] - mx);
            l += p;
            for (size_t i = 0; i < d; ++i) {
                double v = mx4_row_value(v_rows, v_codes, d, t, i);
                o[i] += p * v;
                a[i] += p * fabs(v);
            }
        }
        for (size_t i = 0; i < d; ++i) {
            out[m * d + i] = (l > 0.0) ? (float)(o[i] / l) : 0.0f;
            if (mag) mag[m * d + i] = (l > 0.0) ? (float)(a[i] / l) : 0.0f;
        }
        if (lse) lse[m] = (l > 0.0) ? (float)(mx + log(l)) : -INFINITY;
    }
    free(s); free(o); free(a); free(qd);
}

static size_t compress_int4_g32(const float* xf, size_t n, uint8_t* rec)
{
    size_t groups = n / 32;
    uint8_t* nib = rec + 2 * groups;
    memset(nib, 0, n / 2);
    for (size_t g = 0; g < groups; ++g) {
        float mx = 0.0f;
        for (size_t i = 0; i < 32; ++i) { float a = fabsf(xf[g * 32 + i]); if (a > mx) mx = a; }
        uint16_t s16 = orc_float_to_half(mx / 7.0f);
        memcpy(rec + 2 * g, &s16, 2);
        float s = orc_half_to_float(s16);
        for (size_t i = 0; i < 32; ++i) {
            float r = 0.0f;
            if (s != 0.0f && s == s) {
                r = roundf(xf[g * 32 + i] / s);
                if (!(r == r)) r = 0.0f;
                if (r > 7.0f) r = 7.0f;
                if (r < -7.0f) r = -7.0f;
            }
            uint8_t q = (uint8_t)((int32_t)r & 0xF);
            size_t e = g * 32 + i;
            nib[e >> 1] |= (uint8_t)((e & 1) ? (q << 4) : q);
        }
    }
    return 2 * groups + n / 2;
}

size_t orc_compress_block_f16(const uint16_t* x, size_t n, int scheme, int mode,
                              float* scale, uint8_t* rec)
{
    if (scheme == ORC_COMP_INT4_G32 || scheme == ORC_COMP_FP8_E4M3 || scheme == ORC_COMP_MXFP4) {
        float* xf = (float*)malloc((n ? n : 1) * sizeof(float));
        for (size_t i = 0; i < n; ++i) xf[i] = orc_half_to_float(x[i]);
        size_t len;
        if (scheme == ORC_COMP_INT4_G32) {
            *scale = 1.0f;
            len = compress_int4_g32(xf, n, rec);
        } else if (scheme == ORC_COMP_MXFP4) {
            *scale = 1.0f;
            len = compress_mxfp4(xf, n, rec);
        } else {
            float mx = 0.0f;
            for (size_t i = 0; i < n; ++i) { float a = fabsf(xf[i]); if (a > mx) mx = a; }
            float s = (mx > 0.0f) ? (mx / 448.0f) : 1.0f;
            *scale = s;
            for (size_t i = 0; i < n; ++i) {
                float v = xf[i] / s;
                if (v > 448.0f) v = 448.0f;
                if (v < -448.0f) v = -448.0f;
                rec[i] = orc_f32_to_e4m3(v);
            }
            len = n;
        }
        free(xf);
        return len;
    }
    if (scheme == ORC_COMP_FP16) {
        memcpy(rec, x, 2 * n);
        *scale = 1.0f;
        return 2 * n;
    }
    float* xf = (float*)malloc((n ? n : 1) * sizeof(float));
    for (size_t i = 0; i < n; ++i) xf[i] = orc_half_to_float(x[i]);
    size_t len;
    if (scheme == ORC_COMP_INT8) {
        float s = orc_compute_scale(xf, n);
        *scale = s;
        orc_quantize(xf, n, s, mode, (int8_t*)rec);
        len = n;
    } else {
        len = orc_compress_f32(xf, n, mode, scale, rec);
    }
    free(xf);
    return len;
}

size_t orc_decompress_block_f32(const uint8_t* rec, size_t len, float scale, int scheme,
                                int mode, float* y, size_t cap)
{
    if (scheme == ORC_COMP_INT4_G32) {
        /* fixed-size record: cap/32 group scales then cap/2 nibble bytes; a short record decodes to zeros */
        size_t groups = cap / 32;
        if (len < 2 * groups + cap / 2) { for (size_t i = 0; i < cap; ++i) y[i] = 0.0f; return cap; }
        const uint8_t* nib = rec + 2 * groups;
        for (size_t i = 0; i < cap; ++i) {
            uint16_t s16; memcpy(&s16, rec + 2 * (i / 32), 2);
            uint8_t q4 = (uint8_t)((nib[i >> 1] >> ((i & 1) * 4)) & 0xF);
            int q = (q4 & 8) ? (int)q4 - 16 : (int)q4;
            y[i] = (float)q * orc_half_to_float(s16);
        }
        return cap;
    }
    if (scheme == ORC_COMP_MXFP4) {
        /* fixed-size record: cap/2 nibble bytes then cap/32 E8M0 codes; a short record decodes to zeros */
        size_t groups = cap / 32;
        if (len < cap / 2 + groups) { for (size_t i = 0; i < cap; ++i) y[i] = 0.0f; return cap; }
        const uint8_t* codes = rec + cap / 2;
        for (size_t nib = 0; nib < cap; ++nib) {
            uint8_t nb = (uint8_t)((rec[nib >> 1] >> ((nib & 1) * 4)) & 0xF);
            y[mx4_elem_of_nibble(nib, cap)] = orc_e2m1_to_f32(nb) * orc_e8m0_to_f32(codes[nib / 32]);   /* exact: two significant bits times a power of two */
        }
        return cap;
    }
    if (scheme == ORC_COMP_FP8_E4M3) {
        size_t n = len < cap ? len : cap;
        for (size_t i = 0; i < n; ++i) y[i] = orc_e4m3_to_f32(rec[i]) * scale;
        return n;
    }
    if (scheme == ORC_COMP_FP16) {
        size_t n = len / 2; if (n > cap) n = cap;
        for (size_t i = 0; i < n; ++i) {
            uint16_t h; memcpy(&h, rec + 2 * i, 2);
            y[i] = orc_half_to_float(h);
        }
        return n;
    }
    if (scheme == ORC_COMP_INT8) {
        size_t n = len < cap ? len : cap;
        orc_dequantize((const int8_t*)rec, n, scale, mode, y);
        return n;
    }
    return orc_decompress_f32(rec, len, scale, mode, y, cap);
}

size_t orc_decompress_block_f16(const uint8_t* rec, size_t len, float scale, int scheme,
                                int mode, uint16_t* y, size_t cap)
{
    if (scheme == ORC_COMP_FP16) {
        size_t n = len / 2; if (n > cap) n = cap;
        memcpy(y, rec, 2 * n);
        return n;
    }
    float sy[ORC_STACK_ELEMS];
    float* yf = cap <= ORC_STACK_ELEMS ? sy : (float*)malloc(cap * sizeof(float));
    size_t n = orc_decompress_block_f32(rec, len, scale, scheme, mode, yf, cap);
    for (size_t i = 0; i < n; ++i) y[i] = orc_float_to_half(yf[i]);
    if (yf != sy) free(yf);
    return n;
}

void orc_quantize_rows_e4m3(const uint16_t* q16, size_t rows, size_t d, uint8_t* q8, float* scale)
{
    for (size_t r = 0; r < rows; ++r) {
        float mx = 0.0f;
        for (size_t i = 0; i < d; ++i) { float a = fabsf(orc_half_to_float(q16[r * d + i])); if (a > mx) mx = a; }
        float s = (mx > 0.0f) ? (mx / 448.0f) : 1.0f;
        scale[r] = s;
        for (size_t i = 0; i < d; ++i) {
            float v = orc_half_to_float(q16[r * d + i]) / s;
            if (v > 448.0f) v = 448.0f;
            if (v < -448.0f) v = -448.0f;
            q8[r * d + i] = orc_f32_to_e4m3(v);
        }
    }
}

void orc_qk_scores_fp8(const uint8_t* q8, const float* q_scale, size_t g, const uint8_t* k_rec,
                       const float* k_scale, size_t n_pos, size_t d, float* out)
{
    for (size_t m = 0; m < g; ++m)
        for (size_t t = 0; t < n_pos; ++t) {
            float acc = 0.0f;
            for (size_t i = 0; i < d; ++i)
                acc += orc_e4m3_to_f32(q8[m * d + i]) * orc_e4m3_to_f32(k_rec[t * d + i]);
            out[m * n_pos + t] = acc * k_scale[t] * q_scale[m];
        }
}

void orc_attend_fp8(const uint8_t* q8, const float* q_scale, size_t g, const uint8_t* k_rec, const float* k_scale,
                    const uint8_t* v_rec, const float* v_scale, size_t n_pos, size_t d, float sm_scale,
                    float* out, float* lse, float* mag)
{
    double* s = (double*)malloc((n_pos ? n_pos : 1) * sizeof(double));
    double* o = (double*)malloc(d * sizeof(double));
    double* a = (double*)malloc(d * sizeof(double));
    for (size_t m = 0; m < g; ++m) {
        double mx = -INFINITY;
        for (size_t t = 0; t < n_pos; ++t) {
            double acc = 0.0;
            for (size_t i = 0; i < d; ++i)
                acc += (double)orc_e4m3_to_f32(q8[m * d + i]) * (double)orc_e4m3_to_f32(k_rec[t * d + i]);
            s[t] = acc * (double)k_scale[t] * (double)q_scale[m] * (double)sm_scale;
            if (s[t] > mx) mx = s[t];
        }
        double l = 0.0;
        for (size_t i = 0; i < d; ++i) { o[i] = 0.0; a[i] = 0.0; }
        for (size_t t = 0; t < n_pos; ++t) {
            double p = exp(s[t] - mx);
            l += p;
            for (size_t i = 0; i < d; ++i) {
                double v = (double)orc_e4m3_to_f32(v_rec[t * d + i]) * (double)v_scale[t];
                o[i] += p * v;
                a[i] += p * fabs(v);
            }
        }
        for (size_t i = 0; i < d; ++i) {
            out[m * d + i] = (l > 0.0) ? (float)(o[i] / l) : 0.0f;
            if (mag) mag[m * d + i] = (l > 0.0) ? (float)(a[i] / l) : 0.0f;
        }
        if (lse) lse[m] = (l > 0.0) ? (float)(mx + log(l)) : -INFINITY;
    }
    free(s); free(o); free(a);
}

void orc_attend_f16(const uint16_t* q16, size_t g, const uint16_t* k16, const uint16_t* v16, size_t n_pos, size_t d,
                    float sm_scale, float* out, float* lse, float* mag)
{
    double* s = (double*)malloc((n_pos ? n_pos : 1) * sizeof(double));
    double* o = (double*)malloc(d * sizeof(double));
    double* a = (double*)malloc(d * sizeof(double));
    for (size_t m = 0; m < g; ++m) {
        double mx = -INFINITY;
        for (size_t t = 0; t < n_pos; ++t) {
            double acc = 0.0;
            for (size_t i = 0; i < d; ++i)
                acc += (double)orc_half_to_float(q16[m * d + i]) * (double)orc_half_to_float(k16[t * d + i]);
            s[t] = acc * (double)sm_scale;
            if (s[t] > mx) mx = s[t];
        }
        double l = 0.0;
        for (size_t i = 0; i < d; ++i) { o[i] = 0.0; a[i] = 0.0; }
        for (size_t t = 0; t < n_pos; ++t) {
            double p = exp(s[t] - mx);
            l += p;
            for (size_t i = 0; i < d; ++i) {
                double v = (double)orc_half_to_float(v16[t * d + i]);
                o[i] += p * v;
                a[i] += p * fabs(v);
            }
        }
        for (size_t i = 0; i < d; ++i) {
            out[m * d + i] = (l > 0.0) ? (float)(o[i] / l) : 0.0f;
            if (mag) mag[m * d + i] = (l > 0.0) ? (float)(a[i] / l) : 0.0f;
        }
        if (lse) lse[m] = (l > 0.0) ? (float)(mx + log(l)) : -INFINITY;
    }
    free(s); free(o); free(a);
}

double orc_layer_compression_ratio(uint32_t layer_id)
{   /* cache_engine.cpp:25-33,142-148 : 80/3 = 26, 2*80/3 = 53 */
    if (layer_id >= 80) return 3.2;
    if (layer_id < 80 / 3) return 3.5;
    if (layer_id > 2 * 80 / 3) return 2.75;
    return 3.2;
}
double orc_codec_throughput_gbps(size_t num_engines, double mhz, size_t width_bits)
{   /* cache_engine.cpp:291-296 */
    double per_engine = ((double)width_bits / 8.0) * (mhz / 1000.0);
    return per_engine * (double)num_engines;
}
size_t orc_codec_pipeline_latency_cycles(void) { return 25; } /* cache_engine.cpp:286-289 */

/* ---- TLB ----------------------------------------------------------- */
struct orc_tlb { size_t n; uint64_t* va; uint64_t* pa; uint8_t* valid; };
orc_tlb_t* orc_tlb_new(size_t entries)
{
    orc_tlb_t* t = (orc_tlb_t*)calloc(1, sizeof(*t));
    t->n = entries;
    t->va = (uint64_t*)calloc(entries, 8);
    t->pa = (uint64_t*)calloc(entries, 8);
    t->valid = (uint8_t*)calloc(entries, 1);
    return t;
}
void orc_tlb_delete(orc_tlb_t* t) { if (t) { free(t->va); free(t->pa); free(t->valid); free(t); } }
uint64_t orc_tlb_translate(orc_tlb_t* t, uint64_t va, int* was_hit)
{   /* cache_engine.cpp:118-140 */
    size_t idx = (size_t)((va >> 12) % t->n);
    if (t->valid[idx] && t->va[idx] == (va & ~0xFFFULL)) {
        if (was_hit) *was_hit = 1;
        return t->pa[idx] + (va & 0xFFF);
    }
    if (was_hit) *was_hit = 0;
    uint64_t pa = 0x4000000000ULL + (va & 0xFFFFFFFFFFFFULL);
    t->va[idx] = va & ~0xFFFULL;
    t->pa[idx] = pa & ~0xFFFULL;
    t->valid[idx] = 1;
    return pa;
}

/* ================================================================== */
/* memory manager                                                      */
/* ================================================================== */
typedef struct { uint64_t* v; size_t n, cap; } u64vec;
static void vec_push(u64vec* a, uint64_t x)
{
    if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 16; a->v = (uint64_t*)realloc(a->v, a->cap * 8); }
    a->v[a->n++] = x;
}
static void vec_remove_all(u64vec* a, uint64_t x)
{   /* erase(remove(...)) */
    size_t w = 0;
    for (size_t i = 0; i < a->n; ++i) if (a->v[i] != x) a->v[w++] = a->v[i];
    a->n = w;
}

typedef struct {
    uint64_t va, pa;
    int tier, state;
    uint32_t access_count;
    int is_hot;
    uint32_t layer_id;
    int present;
} orc_page_t;

struct orc_mm {
    uint64_t l1_bytes, l2_bytes, l3_bytes, page_size;
    uint64_t next_va, next_pa_l1, next_pa_l2, next_pa_l3;
    orc_page_t* pages; size_t n_pages, cap_pages;   /* indexed by (va - VA0)/page_size */
    u64vec l1, l2, l3, lru;
    orc_mm_stats_t st;
};
#define ORC_VA0 0x100000000ULL

orc_mm_t* orc_mm_new(uint64_t l1_gb, uint64_t l2_gb, uint64_t l3_gb, uint64_t page_size)
{   /* cxl_memory_manager.cpp:9-24 */
    orc_mm_t* m = (orc_mm_t*)calloc(1, sizeof(*m));
    m->l1_bytes = l1_gb << 30; m->l2_bytes = l2_gb << 30; m->l3_bytes = l3_gb << 30;
    m->page_size = page_size;
    m->next_va = ORC_VA0;
    m->next_pa_l1 = 0x8000000000ULL;
    m->next_pa_l2 = 0x10000000000ULL;
    m->next_pa_l3 = 0x20000000000ULL;
    return m;
}
void orc_mm_delete(orc_mm_t* m)
{
    if (!m) return;
    free(m->pages); free(m->l1.v); free(m->l2.v); free(m->l3.v); free(m->lru.v); free(m);
}
static orc_page_t* mm_exact(orc_mm_t* m, uint64_t page_va)
{
    if (page_va < ORC_VA0) return NULL;
    uint64_t d = page_va - ORC_VA0;
    if (d % m->page_size) return NULL;
    uint64_t idx = d / m->page_size;
    if (idx >= m->n_pages || !m->pages[idx].present) return NULL;
    return &m->pages[idx];
}
static orc_page_t* mm_get_page(orc_mm_t* m, uint64_t va)
{   /* cxl_memory_manager.cpp:279-283 */
    return mm_exact(m, (va / m->page_size) * m->page_size);
}
static u64vec* mm_tier_vec(orc_mm_t* m, int tier)
{
    return tier == ORC_TIER_L1 ? &m->l1 : tier == ORC_TIER_L2 ? &m->l2 : &m->l3;
}
static int mm_can_fit(orc_mm_t* m, int tier, uint64_t size)
{   /* cxl_memory_manager.cpp:295-316 : counts list entries, not pages */
    uint64_t used = (uint64_t)mm_tier_vec(m, tier)->n * m->page_size;
    uint64_t avail = tier == ORC_TIER_L1 ? m->l1_bytes : tier == ORC_TIER_L2 ? m->l2_bytes : m->l3_bytes;
    return used + size <= avail;
}
static void mm_update_lru(orc_mm_t* m, uint64_t va)
{   /* cxl_memory_manager.cpp:318-323 */
    vec_remove_all(&m->lru, va);
    vec_push(&m->lru, va);
}

uint64_t orc_mm_allocate(orc_mm_t* m, uint64_t size_bytes, uint32_t layer_id, int tier)
{   /* cxl_memory_manager.cpp:28-80 */
    uint64_t n = (size_bytes + m->page_size - 1) / m->page_size;
    uint64_t req = n * m->page_size;
    int actual = tier;
    if (tier == ORC_TIER_L1 && !mm_can_fit(m, ORC_TIER_L1, req)) actual = ORC_TIER_L3;
    uint64_t va = m->next_va, pa;
    if (actual == ORC_TIER_L1)      { pa = m->next_pa_l1; m->next_pa_l1 += req; vec_push(&m->l1, va); }
    else if (actual == ORC_TIER_L2) { pa = m->next_pa_l2; m->next_pa_l2 += req; vec_push(&m->l2, va); }
    else                            { pa = m->next_pa_l3; m->next_pa_l3 += req; vec_push(&m->l3, va); }
    uint64_t first = (va - ORC_VA0) / m->page_size;
    if (first + n > m->cap_pages) {
        size_t nc = m->cap_pages ? m->cap_pages : 1024;
        while (nc < first + n) nc *= 2;
        m->pages = (orc_page_t*)realloc(m->pages, nc * sizeof(orc_page_t));
        memset(m->pages + m->cap_pages, 0, (nc - m->cap_pages) * sizeof(orc_page_t));
        m->cap_pages = nc;
    }
    for (uint64_t i = 0; i < n; ++i) {
        orc_page_t* p = &m->pages[first + i];
        p->va = va + i * m->page_size;
        p->pa = pa + i * m->page_size;
        p->tier = actual; p->state = ORC_STATE_EXCLUSIVE;
        p->access_count = 0; p->is_hot = 0; p->layer_id = layer_id; p->present = 1;
    }
    if (first + n > m->n_pages) m->n_pages = first + n;
    m->next_va += req;
    return va;
}

void orc_mm_deallocate(orc_mm_t* m, uint64_t va)
{   /* cxl_memory_manager.cpp:82-104 : exact-key lookup, erases ONE page */
    orc_page_t* p = mm_exact(m, va);
    if (!p) return;
    if (p->tier == ORC_TIER_L1) { vec_remove_all(&m->l1, va); vec_remove_all(&m->lru, va); }
    else vec_remove_all(mm_tier_vec(m, p->tier), va);
    p->present = 0;
}

uint64_t orc_mm_translate(orc_mm_t* m, uint64_t va)
{   /* cxl_memory_manager.cpp:106-117 */
    uint64_t page_va = (va / m->page_size) * m->page_size;
    orc_page_t* p = mm_exact(m, page_va);
    return p ? p->pa + (va - page_va) : 0;
}
int orc_mm_is_in_cache(orc_mm_t* m, uint64_t va, int tier)
{   /* cxl_memory_manager.cpp:119-128 */
    orc_page_t* p = mm_get_page(m, va);
    return p ? p->tier == tier : 0;
}

int orc_mm_demote_to_l3(orc_mm_t* m, uint64_t va)
{   /* cxl_memory_manager.cpp:165-194 */
    orc_page_t* p = mm_get_page(m, va);
    if (!p || p->tier == ORC_TIER_L3) return 0;
    int old = p->tier;
    p->tier = ORC_TIER_L3;
    if (old == ORC_TIER_L1) {
        vec_remove_all(&m->l1, va); vec_remove_all(&m->lru, va);
        m->st.migrations_l1_to_l3++;
    } else if (old == ORC_TIER_L2) {
        vec_remove_all(&m->l2, va);
    }
    vec_push(&m->l3, va);
    return 1;
}

int orc_mm_promote_to_l1(orc_mm_t* m, uint64_t va)
{   /* cxl_memory_manager.cpp:130-163.  The reference self-deadlocks when it
     * has to evict; the restatement performs the eviction the
     * code intends (front of the LRU list goes to L3). */
    orc_page_t* p = mm_get_page(m, va);
    if (!p || p->tier == ORC_TIER_L1) return 0;
    if (!mm_can_fit(m, ORC_TIER_L1, m->page_size) && m->lru.n) {
        uint64_t victim = m->lru.v[0];
        memmove(m->lru.v, m->lru.v + 1, (m->lru.n - 1) * 8); m->lru.n--;
        orc_mm_demote_to_l3(m, victim);
    }
    int old = p->tier;
    p->tier = ORC_TIER_L1;
    if (old == ORC_TIER_L2) vec_remove_all(&m->l2, va);
    else if (old == ORC_TIER_L3) { vec_remove_all(&m->l3, va); m->st.migrations_l3_to_l1++; }
    vec_push(&m->l1, va);
    mm_update_lru(m, va);
    return 1;
}

void orc_mm_invalidate_page(orc_mm_t* m, uint64_t va)
{ orc_page_t* p = mm_get_page(m, va); if (p) p->state = ORC_STATE_INVALID; }
void orc_mm_mark_modified(orc_mm_t* m, uint64_t va)
{ orc_page_t* p = mm_get_page(m, va); if (p) p->state = ORC_STATE_MODIFIED; }
int orc_mm_get_page_state(orc_mm_t* m, uint64_t va)
{ orc_page_t* p = mm_get_page(m, va); return p ? p->state : ORC_STATE_INVALID; }

void orc_mm_update_access_tracking(orc_mm_t* m, uint64_t va)
{   /* cxl_memory_manager.cpp:223-245 */
    orc_page_t* p = mm_get_page(m, va);
    if (!p) return;
    p->access_count++;
    if (p->tier == ORC_TIER_L1) m->st.l1_hits++;
    else if (p->tier == ORC_TIER_L2) m->st.l2_hits++;
    else m->st.l3_accesses++;
    mm_update_lru(m, va);
}
int orc_mm_is_hot_page(orc_mm_t* m, uint64_t va)
{   /* cxl_memory_manager.cpp:247-257 */
    orc_page_t* p = mm_get_page(m, va);
    if (!p) return 0;
    p->is_hot = p->access_count > 10;
    return p->is_hot;
}
void orc_mm_get_statistics(orc_mm_t* m, orc_mm_stats_t* out)
{   /* cxl_memory_manager.cpp:259-274 */
    *out = m->st;
    uint64_t t1 = out->l1_hits + out->l1_misses, t2 = out->l2_hits + out->l2_misses;
    if (t1) out->l1_hit_rate = (double)out->l1_hits / (double)t1;
    if (t2) out->l2_hit_rate = (double)out->l2_hits / (double)t2;
}

uint64_t orc_mm_cxl_access(orc_mm_t* m, uint64_t base_va, uint64_t offset)
{   /* memory_allocator.cpp:105-143 */
    uint64_t va = base_va + offset;
    orc_mm_update_access_tracking(m, va);
    if (orc_mm_is_in_cache(m, va, ORC_TIER_L1)) return va;
    if (orc_mm_is_in_cache(m, va, ORC_TIER_L2)) {
        if (orc_mm_is_hot_page(m, va)) orc_mm_promote_to_l1(m, va);
        return va;
    }
    orc_mm_promote_to_l1(m, va);
    return va;
}

/* ================================================================== */
/* prefetcher                                                          */
/* ================================================================== */
uint64_t orc_compute_kv_address(uint32_t req_id, uint32_t layer_id, uint32_t position)
{   /* speculative_prefetcher.cpp:153-160 */
    return ((uint64_t)req_id << 32) | ((uint64_t)layer_id << 16) | (uint64_t)position;
}

size_t orc_prefetch_legacy(orc_mm_t* mm, uint32_t layer_id, size_t depth,
                           size_t n_predictions, uint64_t* out_va)
{   /* speculative_prefetcher.cpp:35-67 : req_id fixed 0, position i+1 */
    (void)depth;
    size_t w = 0;
    for (size_t i = 0; i < n_predictions; ++i) {
        uint64_t va = orc_compute_kv_address(0, layer_id, (uint32_t)(i + 1));
        if (mm && (orc_mm_is_in_cache(mm, va, ORC_TIER_L1) ||
                   orc_mm_is_in_cache(mm, va, ORC_TIER_L2)))
            continue;
        out_va[w++] = va;
    }
    return w;
}

struct orc_adapt { size_t depth; double hist[100]; size_t n; };
orc_adapt_t* orc_adapt_new(size_t d)
{ orc_adapt_t* a = (orc_adapt_t*)calloc(1, sizeof(*a)); a->depth = d; return a; }
void orc_adapt_delete(orc_adapt_t* a) { free(a); }
void orc_adapt_update(orc_adapt_t* a, int ok)
{   /* speculative_prefetcher.cpp:98-120 : window 100, decision on last 10 */
    if (a->n == 100) { memmove(a->hist, a->hist + 1, 99 * sizeof(double)); a->n = 99; }
    a->hist[a->n++] = ok ? 1.0 : 0.0;
    if (a->n >= 10) {
        double acc = 0.0;
        for (size_t i = a->n - 10; i < a->n; ++i) acc += a->hist[i];
        acc /= 10.0;
        if (acc > 0.95 && a->depth < 8) a->depth++;
        else if (acc < 0.85 && a->depth > 2) a->depth--;
    }
}
size_t orc_adapt_depth(const orc_adapt_t* a) { return a->depth; }
int orc_is_misprediction(uint32_t actual, const uint32_t* predicted, size_t n)
{   /* speculative_prefetcher.cpp:84-96 */
    for (size_t i = 0; i < n; ++i) if (predicted[i] == actual) return 0;
    return 1;
}

void orc_lstm_reference_weights(unsigned seed, size_t vocab, size_t emb_dim, size_t hidden,
                                size_t layers, float* emb, float* wout)
{   /* lstm_predictor.cpp:22-35 */
    srand(seed);
    for (size_t i = 0; i < vocab * emb_dim; ++i) emb[i] = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    for (size_t i = 0; i < layers * hidden * hidden * 4; ++i) (void)rand();
    for (size_t i = 0; i < hidden * vocab; ++i) wout[i] = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
}

size_t orc_lstm_predict(const float* emb, const float* wout, size_t vocab, size_t emb_dim,
                        size_t hidden, size_t layers, size_t hist_len,
                        const uint32_t* history, size_t n_hist, size_t k,
                        uint32_t* out_tok, float* out_prob)
{
    /* lstm_predictor.cpp:44-51 : last hist_len tokens, zero-padded at the front */
    uint32_t* h = (uint32_t*)calloc(hist_len ? hist_len : 1, sizeof(uint32_t));
    if (n_hist >= hist_len) memcpy(h, history + (n_hist - hist_len), hist_len * sizeof(uint32_t));
    else memcpy(h + (hist_len - n_hist), history, n_hist * sizeof(uint32_t));
    float* hid = (float*)calloc(hidden, sizeof(float));
    float* cell = (float*)calloc(hidden, sizeof(float));
    float* e = (float*)calloc(emb_dim, sizeof(float));
    for (size_t t = 0; t < hist_len; ++t) {
        /* embed_token, lstm_predictor.cpp:148-159 */
        for (size_t j = 0; j < emb_dim; ++j) e[j] = (h[t] < vocab) ? emb[(size_t)h[t] * emb_dim + j] : 0.0f;
        for (size_t l = 0; l < layers; ++l) {
            /* lstm_forward, lstm_predictor.cpp:117-146 */
            for (size_t i = 0; i < hidden; ++i) {
                float g = 0.0f;
                for (size_t j = 0; j < emb_dim && j < hidden; ++j) g += e[j] * 0.1f;
                cell[i] = 0.5f * cell[i] + 0.5f * tanhf(g);
                hid[i] = 0.5f * tanhf(cell[i]);
            }
        }
    }
    /* compute_output_probs, lstm_predictor.cpp:161-188 */
    float* p = (float*)calloc(vocab ? vocab : 1, sizeof(float));
    for (size_t i = 0; i < vocab; ++i)
        for (size_t j = 0; j < hidden; ++j) p[i] += hid[j] * wout[i * hidden + j];
    float mx = p[0];
    for (size_t i = 1; i < vocab; ++i) if (p[i] > mx) mx = p[i];
    float sum = 0.0f;
    for (size_t i = 0; i < vocab; ++i) { p[i] = expf(p[i] - mx); sum += p[i]; }
    for (size_t i = 0; i < vocab; ++i) p[i] /= sum;
    /* top-k by probability (lstm_predictor.cpp:75-93) */
    size_t n = k < vocab ? k : vocab;
    for (size_t r = 0; r < n; ++r) {
        size_t best = 0; float bp = -1.0f;
        for (size_t i = 0; i < vocab; ++i) if (p[i] > bp) { bp = p[i]; best = i; }
        out_tok[r] = (uint32_t)best; out_prob[r] = bp; p[best] = -2.0f;
    }
    free(h); free(hid); free(cell); free(e); free(p);
    return n;
}

uint64_t orc_rtl_prefetch_vaddr(uint32_t req_id, uint16_t layer, uint32_t pos_plus)
{   /* prefetch_core.v:92-98 : the 89-bit concatenation keeps its low 64 bits:
     * bit0 kind=0, bits 1..32 pos, bits 33..40 head=0, bits 41..56 layer,
     * bits 57..63 req[6:0]. */
    return ((uint64_t)pos_plus << 1) | ((uint64_t)layer << 41) | ((uint64_t)req_id << 57);
}

size_t orc_prefetch_pages(uint32_t req_id, uint32_t layer, uint32_t cur_pos, uint32_t depth_k,
                          uint64_t L, uint64_t T, uint64_t H, uint64_t D, uint64_t bpe,
                          uint64_t alloc_pages, const uint32_t* flags,
                          uint64_t* out, size_t cap)
{
    size_t w = 0;
    uint64_t entry = D * bpe, row = H * entry;
    if (row == 0) return 0;
    for (uint64_t kind = 0; kind < 2; ++kind) {
        int have_last = 0; uint64_t last = 0;
        for (uint64_t i = 1; i <= depth_k; ++i) {
            uint64_t p = (uint64_t)cur_pos + i;
            if (p >= T) break;
            uint64_t off = orc_calc_offset(req_id, layer, 0, p, kind, entry, L, T, H);
            uint64_t pg0 = off / ORC_PAGE_SIZE, pg1 = (off + row - 1) / ORC_PAGE_SIZE;
            for (uint64_t pg = pg0; pg <= pg1; ++pg) {
                if (have_last && pg <= last) continue;
                have_last = 1; last = pg;
                if (pg >= alloc_pages) continue;
                if (flags && (flags[pg] & 0x3u)) continue;
                if (w < cap) out[w] = pg;
                w++;
            }
        }
    }
    return w < cap ? w : cap;
}

/* ---- coherence shadow directory (coherence_manager.cpp) --------------------- */
typedef struct { uint64_t line; uint8_t used, state, tier; uint32_t access_count; } coh_entry;
struct orc_coh { size_t line_size; int has_driver; coh_entry* tab; size_t cap, n; uint64_t st[7]; };
enum { COH_READS, COH_WRITES, COH_OPS, COH_INVAL, COH_WB, COH_HITS, COH_MISSES };
enum { COH_OP_READ, COH_OP_WRITE, COH_OP_INVALIDATE, COH_OP_WRITEBACK };

orc_coh_t* orc_coh_new(size_t cache_line_size, int has_driver)
{
    orc_coh_t* c = (orc_coh_t*)calloc(1, sizeof(*c));
    c->line_size = cache_line_size;
    c->has_driver = has_driver;
    c->cap = 1024;
    c->tab = (coh_entry*)calloc(c->cap, sizeof(coh_entry));
    return c;
}
void orc_coh_delete(orc_coh_t* c) { if (c) { free(c->tab); free(c); } }
static uint64_t coh_align(const orc_coh_t* c, uint64_t a) { return a & ~((uint64_t)c->line_size - 1u); }  /* coherence_manager.h:206-208 */
static size_t coh_slot(const orc_coh_t* c, uint64_t line) { return (size_t)((line * 0x9E3779B97F4A7C15ull) >> 20) & (c->cap - 1); }
static coh_entry* coh_find(const orc_coh_t* c, uint64_t line)
{
    for (size_t i = coh_slot(c, line);; i = (i + 1) & (c->cap - 1)) {
        if (!c->tab[i].used) return NULL;
        if (c->tab[i].line == line) return &c->tab[i];
    }
}
static coh_entry* coh_get_or_create(orc_coh_t* c, uint64_t line)
{   /* coherence_manager.cpp:384-396: new entries are INVALID, L3_CXL, access_count 0 */
    coh_entry* e = coh_find(c, line);
    if (e) return e;
    if ((c->n + 1) * 2 > c->cap) {
        coh_entry* old = c->tab; size_t oc = c->cap;
        c->cap *= 2;
        c->tab = (coh_entry*)calloc(c->cap, sizeof(coh_entry));
        for (size_t i = 0; i < oc; ++i)
            if (old[i].used) {
                size_t j = coh_slot(c, old[i].line);
                while (c->tab[j].used) j = (j + 1) & (c->cap - 1);
                c->tab[j] = old[i];
            }
        free(old);
    }
    size_t i = coh_slot(c, line);
    while (c->tab[i].used) i = (i + 1) & (c->cap - 1);
    c->tab[i].used = 1; c->tab[i].line = line; c->tab[i].state = 0; c->tab[i].tier = 2; c->tab[i].access_count = 0;
    c->n++;
    return &c->tab[i];
}
static void coh_count(orc_coh_t* c, int op, int hit)
{   /* coherence_manager.cpp:436-458 */
    if (op == COH_OP_READ) { c->st[COH_READS]++; c->st[hit ? COH_HITS : COH_MISSES]++; }
    else if (op == COH_OP_WRITE) { c->st[COH_WRITES]++; c->st[hit ? COH_HITS : COH_MISSES]++; }
    else c->st[COH_OPS]++;
}
static int coh_send(orc_coh_t* c, int op)
{   /* coherence_manager.cpp:398-424: no driver -> false (nothing counted); else counted as a hit, true */
    if (!c->has_driver) return 0;
    coh_count(c, op, 1);
    return 1;
}
int orc_coh_request_read(orc_coh_t* c, uint64_t addr)
{   /* :33-68 */
    uint64_t line = coh_align(c, addr);
    coh_entry* e = coh_find(c, line);
    if (e && e->state != 0) { coh_count(c, COH_OP_READ, 1); e->access_count++; return 1; }
    coh_count(c, COH_OP_READ, 0);
    int ok = coh_send(c, COH_OP_READ);
    if (ok) { e = coh_get_or_create(c, line); e->state = 1; e->tier = 0; e->access_count = 1; }
    return ok;
}
int orc_coh_request_write(orc_coh_t* c, uint64_t addr)
{   /* :70-106 */
    uint64_t line = coh_align(c, addr);
    coh_entry* e = coh_find(c, line);
    if (e && e->state == 1) { coh_count(c, COH_OP_INVALIDATE, 0); c->st[COH_INVAL]++; }
    coh_count(c, COH_OP_WRITE, e != NULL);
    int ok = coh_send(c, COH_OP_WRITE);
    if (ok) { e = coh_get_or_create(c, line); e->state = 3; e->tier = 0; e->access_count++; }
    return ok;
}
int orc_coh_invalidate(orc_coh_t* c, uint64_t addr)
{   /* :108-134 */
    coh_entry* e = coh_find(c, coh_align(c, addr));
    if (!e) return 1;
    if (e->state == 3) c->st[COH_WB]++;
    e->state = 0;
    int ok = coh_send(c, COH_OP_INVALIDATE);
    c->st[COH_INVAL]++;
    return ok;
}
int orc_coh_writeback(orc_coh_t* c, uint64_t addr)
{   /* :136-158 */
    coh_entry* e = coh_find(c, coh_align(c, addr));
    if (!e || e->state != 3) return 1;
    int ok = coh_send(c, COH_OP_WRITEBACK);
    if (ok) { e->state = 1; e->tier = 2; c->st[COH_WB]++; }
    return ok;
}
int orc_coh_flush_all(orc_coh_t* c)
{   /* :160-181 (the send result is ignored there) */
    uint64_t flushed = 0;
    for (size_t i = 0; i < c->cap; ++i)
        if (c->tab[i].used && c->tab[i].state == 3) { (void)coh_send(c, COH_OP_WRITEBACK); c->tab[i].state = 1; c->tab[i].tier = 2; flushed++; }
    c->st[COH_WB] += flushed;
    return 1;
}
int orc_coh_get_state(const orc_coh_t* c, uint64_t addr) { coh_entry* e = coh_find(c, coh_align(c, addr)); return e ? e->state : 0; }
int orc_coh_get_tier(const orc_coh_t* c, uint64_t addr) { coh_entry* e = coh_find(c, coh_align(c, addr)); return e ? e->tier : 2; }
int orc_coh_promote_to_l1(orc_coh_t* c, uint64_t addr)
{   /* :211-238: creates the entry (INVALID) when absent */
    coh_entry* e = coh_get_or_create(c, coh_align(c, addr));
    if (e->tier == 0) return 1;
    int ok = coh_send(c, COH_OP_READ);
    if (ok) e->tier = 0;
    return ok;
}
int orc_coh_demote_to_l3(orc_coh_t* c, uint64_t addr)
{   /* :240-261 */
    coh_entry* e = coh_find(c, coh_align(c, addr));
    if (!e || e->tier == 2) return 1;
    if (e->state == 3) { (void)coh_send(c, COH_OP_WRITEBACK); e->state = 1; c->st[COH_WB]++; }
    e->tier = 2;
    return 1;
}
void orc_coh_update_tier(orc_coh_t* c, uint64_t addr, int tier) { coh_get_or_create(c, coh_align(c, addr))->tier = (uint8_t)tier; }   /* :263-270 */
int orc_coh_batch_invalidate(orc_coh_t* c, const uint64_t* addrs, size_t n)
{   /* :272-289: only present lines get an operation, every address counts as an invalidation */
    int all = 1;
    for (size_t i = 0; i < n; ++i) {
        coh_entry* e = coh_find(c, coh_align(c, addrs[i]));
        if (e) { e->state = 0; all &= coh_send(c, COH_OP_INVALIDATE); }
    }
    c->st[COH_INVAL] += n;
    return all;
}
void orc_coh_get_statistics(const orc_coh_t* c, uint64_t* stats7) { memcpy(stats7, c->st, sizeof(c->st)); }
void orc_coh_reset_statistics(orc_coh_t* c) { memset(c->st, 0, sizeof(c->st)); }

/* ================================================================== */
/* batch drivers: the per-block functions above over many blocks, on   */
/* several threads (blocks are independent).  Test convenience only:   */
/* the arithmetic is orc_compress_block_f16 / orc_decompress_block_f16 */
/* ================================================================== */
#include <pthread.h>

typedef struct {
    const uint16_t* x; uint8_t* recs; size_t rec_stride; uint32_t* lens; float* scales; uint16_t* y;
    size_t n, b0, b1; int scheme, mode, compress;
} blocks_job;

static void* blocks_worker(void* arg)
{
    blocks_job* j = (blocks_job*)arg;
    for (size_t b = j->b0; b < j->b1; ++b) {
        if (j->compress) {
            float s = 1.0f;
            j->lens[b] = (uint32_t)orc_compress_block_f16(j->x + b * j->n, j->n, j->scheme, j->mode, &s, j->recs + b * j->rec_stride);
            j->scales[b] = s;
        } else {
            size_t got = orc_decompress_block_f16(j->recs + b * j->rec_stride, j->lens[b], j->scales[b], j->scheme, j->mode,
                                                  j->y + b * j->n, j->n);
            for (size_t i = got; i < j->n; ++i) j->y[b * j->n + i] = 0;
        }
    }
    return NULL;
}

static void blocks_run(blocks_job proto, size_t n_blocks, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pthread_t tid[64];
    int joinable[64];
    blocks_job jobs[64];
    const size_t per = (n_blocks + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; ++t) {
        joinable[t] = 0;
        jobs[t] = proto;
        jobs[t].b0 = (size_t)t * per < n_blocks ? (size_t)t * per : n_blocks;
        jobs[t].b1 = jobs[t].b0 + per < n_blocks ? jobs[t].b0 + per : n_blocks;
        if (jobs[t].b0 >= jobs[t].b1) continue;
        /* the last share runs on the calling thread; so does any share whose thread cannot be created */
        if (t + 1 < threads && pthread_create(&tid[t], NULL, blocks_worker, &jobs[t]) == 0) joinable[t] = 1;
        else blocks_worker(&jobs[t]);
    }
    for (int t = 0; t < threads; ++t)
        if (joinable[t]) pthread_join(tid[t], NULL);
}

void orc_compress_blocks_f16(const uint16_t* x, size_t n_blocks, size_t n, int scheme, int mode,
                             float* scales, uint32_t* lens, uint8_t* recs, size_t rec_stride, int threads)
{
    blocks_job j; memset(&j, 0, sizeof(j));
    j.x = x; j.recs = recs; j.rec_stride = rec_stride; j.lens = lens; j.scales = scales; j.n = n;
    j.scheme = scheme; j.mode = mode; j.compress = 1;
    blocks_run(j, n_blocks, threads);
}

void orc_decompress_blocks_f16(const uint8_t* recs, size_t rec_stride, const uint32_t* lens, const float* scales,
                               size_t n_blocks, size_t n, int scheme, int mode, uint16_t* y, int threads)
{
    blocks_job j; memset(&j, 0, sizeof(j));
    j.recs = (uint8_t*)recs; j.rec_stride = rec_stride; j.lens = (uint32_t*)lens; j.scales = (float*)scales; j.y = y; j.n = n;
    j.scheme = scheme; j.mode = mode; j.compress = 0;
    blocks_run(j, n_blocks, threads);
}
