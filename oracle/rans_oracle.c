/* oracle/rans_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C restatement of the entropy-coder algorithms the reference reaches through the
 * third-party wheel CompressAI 1.2.4 (pyproject.toml:16; source NOT under /root/reference, not
 * installable offline): `compressai.ans.RansEncoder.encode_with_indexes`,
 * `RansDecoder.set_stream/decode_stream/decode_with_indexes` and
 * `compressai._CXX.pmf_to_quantized_cdf`.  Reference call sites:
 *   hyperprior_charm_dc_vic_model.py:68,84; minnen20_charm_context_model.py:165,179-180,200-202;
 *   hyperprior_dc_vic_model.py:66-68.
 * Published algorithm restated from SURVEY.md Appendix B (ryg_rans rans64: 64-bit state, 32-bit
 * words, 16-bit probability precision, 4-bit bypass digits for out-of-range symbols, symbols
 * pushed in reverse).  PARITY UNPINNED: the reference holds no golden vector for this boundary;
 * this file is validated by round trips / invariants only (tests/test_entropy_oracle.py).
 *
 * Deliberately simple (linear CDF search, explicit symbol stack).  The product's coder lives in
 * dc_vic_amd/csrc/host_entropy.cpp and shares no code with this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRECISION 16
#define BYPASS_PRECISION 4
#define MAX_BYPASS_VAL 15
#define RANS64_L (1ull << 31)

typedef struct { uint16_t start, range; uint8_t bypass; } sym_t;

/* ---- pmf -> 16-bit quantised CDF (App-B "_pmf_to_cdf"/"pmf_to_quantized_cdf") ---------------- */
/* pmf: n floats (the tail mass already appended by the caller); cdf_out: n+1 uint32.           */
int oracle_pmf_to_quantized_cdf(const float *pmf, int n, uint32_t *cdf) {
    uint32_t total = 0;
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (!(pmf[i] >= 0.0f) || !isfinite(pmf[i])) return -1;
        cdf[i + 1] = (uint32_t)roundf(pmf[i] * (float)(1 << PRECISION));
    }
    for (int i = 0; i <= n; ++i) total += cdf[i];
    if (total == 0) return -2;
    for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)((((uint64_t)1 << PRECISION) * cdf[i]) / total);
    for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
    cdf[n] = 1u << PRECISION;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            uint32_t best_freq = ~0u;
            int best = -1;
            for (int j = 0; j < n; ++j) {
                uint32_t f = cdf[j + 1] - cdf[j];
                if (f > 1 && f < best_freq) { best_freq = f; best = j; }
            }
            if (best < 0) return -3;
            if (best < i) { for (int j = best + 1; j <= i; ++j) cdf[j]--; }
            else          { for (int j = i + 1; j <= best; ++j) cdf[j]++; }
        }
    }
    return 0;
}

/* ---- encoder ------------------------------------------------------------------------------- */
/* cdfs: [n_cdf][cdf_stride] int32 (zero padded), cdf_sizes[n_cdf], offsets[n_cdf].
 * Returns the number of bytes written to out (capacity out_cap), or <0 on error.               */
long oracle_rans_encode(const int32_t *symbols, const int32_t *indexes, long n,
                        const int32_t *cdfs, int cdf_stride, const int32_t *cdf_sizes,
                        const int32_t *offsets, uint8_t *out, long out_cap) {
    long cap = n * 2 + 64, ns = 0;
    sym_t *st = (sym_t *)malloc(sizeof(sym_t) * (size_t)cap);
    if (!st) return -1;
#define PUSH(a, b, c) do { if (ns == cap) { cap *= 2; st = (sym_t *)realloc(st, sizeof(sym_t) * (size_t)cap); } \
        st[ns].start = (uint16_t)(a); st[ns].range = (uint16_t)(b); st[ns].bypass = (c); ++ns; } while (0)
    for (long i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        const int32_t *cdf = cdfs + (long)ci * cdf_stride;
        const int32_t max_value = cdf_sizes[ci] - 2;
        int32_t value = symbols[i] - offsets[ci];
        uint32_t raw = 0;
        if (value < 0) { raw = (uint32_t)(-2 * value - 1); value = max_value; }
        else if (value >= max_value) { raw = (uint32_t)(2 * (value - max_value)); value = max_value; }
        PUSH(cdf[value], cdf[value + 1] - cdf[value], 0);
        if (value == max_value) {
            int32_t nb = 0;
            while ((raw >> (nb * BYPASS_PRECISION)) != 0) ++nb;
            int32_t v = nb;
            while (v >= MAX_BYPASS_VAL) { PUSH(MAX_BYPASS_VAL, MAX_BYPASS_VAL + 1, 1); v -= MAX_BYPASS_VAL; }
            PUSH(v, v + 1, 1);
            for (int32_t j = 0; j < nb; ++j) {
                int32_t d = (raw >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL;
                PUSH(d, d + 1, 1);
            }
        }
    }
    /* flush: pop in reverse, write 32-bit words downwards from the end of a scratch buffer */
    long nwords = ns + 4;
    uint32_t *buf = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nwords);
    uint32_t *ptr = buf + nwords;
    uint64_t x = RANS64_L;
    for (long i = ns - 1; i >= 0; --i) {
        if (!st[i].bypass) {
            uint32_t freq = st[i].range, start = st[i].start;
            uint64_t x_max = ((RANS64_L >> PRECISION) << 32) * freq;
            if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
            x = ((x / freq) << PRECISION) + (x % freq) + start;
        } else {
            uint32_t freq = 1u << (16 - BYPASS_PRECISION);
            uint64_t x_max = ((RANS64_L >> 16) << 32) * freq;
            if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
            x = (x << BYPASS_PRECISION) | st[i].start;
        }
    }
    ptr -= 2;
    ptr[0] = (uint32_t)x;
    ptr[1] = (uint32_t)(x >> 32);
    long nbytes = (long)((buf + nwords) - ptr) * 4;
    long ret = nbytes;
    if (nbytes > out_cap) ret = -2; else memcpy(out, ptr, (size_t)nbytes);
    free(buf); free(st);
    return ret;
}

/* ---- decoder (stateful so that CHARM can pull one slice at a time) -------------------------- */
typedef struct { uint64_t x; const uint32_t *ptr; const uint32_t *end; uint32_t *own; } dec_t;

void *oracle_rans_dec_new(const uint8_t *stream, long nbytes) {
    dec_t *d = (dec_t *)calloc(1, sizeof(dec_t));
    long nw = nbytes / 4;
    d->own = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(nw + 4));
    memset(d->own, 0, sizeof(uint32_t) * (size_t)(nw + 4));
    memcpy(d->own, stream, (size_t)nw * 4);
    d->ptr = d->own; d->end = d->own + nw + 4;
    d->x = (uint64_t)d->ptr[0] | ((uint64_t)d->ptr[1] << 32);
    d->ptr += 2;
    return d;
}

void oracle_rans_dec_free(void *h) { dec_t *d = (dec_t *)h; if (d) { free(d->own); free(d); } }

static uint32_t get_bits(dec_t *d, uint32_t nbits) {
    uint64_t x = d->x;
    uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
    x >>= nbits;
    if (x < RANS64_L && d->ptr < d->end) { x = (x << 32) | *d->ptr; d->ptr += 1; }
    d->x = x;
    return val;
}

int oracle_rans_dec_decode(void *h, const int32_t *indexes, long n, const int32_t *cdfs, int cdf_stride,
                           const int32_t *cdf_sizes, const int32_t *offsets, int32_t *out) {
    dec_t *d = (dec_t *)h;
    for (long i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        const int32_t *cdf = cdfs + (long)ci * cdf_stride;
        const int32_t max_value = cdf_sizes[ci] - 2;
        const uint32_t cum = (uint32_t)(d->x & ((1u << PRECISION) - 1));
        int32_t s = 0;
        while (s < cdf_sizes[ci] && !((uint32_t)cdf[s] > cum)) ++s;   /* first entry > cum */
        s -= 1;
        if (s < 0 || s > max_value) return -1;
        {
            uint64_t x = d->x;
            uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
            x = freq * (x >> PRECISION) + (x & ((1ull << PRECISION) - 1)) - start;
            if (x < RANS64_L && d->ptr < d->end) { x = (x << 32) | *d->ptr; d->ptr += 1; }
            d->x = x;
        }
        int32_t value = s;
        if (value == max_value) {
            int32_t val = (int32_t)get_bits(d, BYPASS_PRECISION);
            int32_t nb = val;
            while (val == MAX_BYPASS_VAL) { val = (int32_t)get_bits(d, BYPASS_PRECISION); nb += val; }
            int32_t raw = 0;
            for (int32_t j = 0; j < nb; ++j) { val = (int32_t)get_bits(d, BYPASS_PRECISION); raw |= val << (j * BYPASS_PRECISION); }
            value = raw >> 1;
            if (raw & 1) value = -value - 1; else value += max_value;
        }
        out[i] = value + offsets[ci];
    }
    return 0;
}
