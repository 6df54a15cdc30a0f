// qual_pack.cpp -- the host half of the pass-bit form of the pileup: "quality >= min_base_quality" (mod.rs:33) taken
// once, where the quality bytes are touched anyway (cl_push_reads' walk), and packed one bit per base; and the sum of
// the passing qualities of a stretch of bases (contig_profiler.rs:68-70: summed_baseq is a per-read separable sum,
// SURVEY 8a-7).  AVX2 when the CPU has it, SSE2 (x86-64 baseline) otherwise; both are checked against the scalar
// form by tests/test_qual_rows.py.
#include "qual_pack.h"

#include <immintrin.h>
#include <string.h>

#include <algorithm>

namespace dut {

namespace {

inline uint64_t mask64_scalar(const uint8_t *q, uint32_t n, uint8_t thr)
{
    uint64_t m = 0;
    for (uint32_t i = 0; i < n; ++i) m |= (uint64_t)(q[i] >= thr) << i;
    return m;
}

inline uint64_t sum_scalar(const uint8_t *q, uint64_t n, uint8_t thr)
{
    uint64_t s = 0;
    for (uint64_t i = 0; i < n; ++i) s += q[i] >= thr ? q[i] : 0u;
    return s;
}

// ---- SSE2 ----
inline uint64_t mask64_sse2(const uint8_t *q, __m128i t)
{
    uint64_t m = 0;
    for (int k = 0; k < 4; ++k) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(q + 16 * k));
        // unsigned v >= t  <=>  max(v, t) == v
        m |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_max_epu8(v, t), v)) << (16 * k);
    }
    return m;
}

void words_sse2(const uint8_t *q, uint64_t n_words, uint8_t thr, uint64_t *out)
{
    const __m128i t = _mm_set1_epi8((char)thr);
    for (uint64_t w = 0; w < n_words; ++w) out[w] = mask64_sse2(q + 64 * w, t);
}

uint64_t sum_sse2(const uint8_t *q, uint64_t n, uint8_t thr)
{
    const __m128i t = _mm_set1_epi8((char)thr), z = _mm_setzero_si128();
    __m128i acc = z;
    uint64_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(q + i));
        const __m128i m = _mm_cmpeq_epi8(_mm_max_epu8(v, t), v);
        acc = _mm_add_epi64(acc, _mm_sad_epu8(_mm_and_si128(v, m), z));
    }
    uint64_t lanes[2];
    _mm_storeu_si128(reinterpret_cast<__m128i *>(lanes), acc);
    return lanes[0] + lanes[1] + sum_scalar(q + i, n - i, thr);
}

// ---- AVX2 ----
__attribute__((target("avx2"))) void words_avx2(const uint8_t *q, uint64_t n_words, uint8_t thr, uint64_t *out)
{
    const __m256i t = _mm256_set1_epi8((char)thr);
    for (uint64_t w = 0; w < n_words; ++w) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(q + 64 * w));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(q + 64 * w + 32));
        const uint32_t ma = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_max_epu8(a, t), a));
        const uint32_t mb = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_max_epu8(b, t), b));
        out[w] = (uint64_t)ma | ((uint64_t)mb << 32);
    }
}

__attribute__((target("avx2"))) uint64_t sum_avx2(const uint8_t *q, uint64_t n, uint8_t thr)
{
    const __m256i t = _mm256_set1_epi8((char)thr), z = _mm256_setzero_si256();
    __m256i acc = z;
    uint64_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(q + i));
        const __m256i m = _mm256_cmpeq_epi8(_mm256_max_epu8(v, t), v);
        acc = _mm256_add_epi64(acc, _mm256_sad_epu8(_mm256_and_si256(v, m), z));
    }
    uint64_t lanes[4];
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(lanes), acc);
    return lanes[0] + lanes[1] + lanes[2] + lanes[3] + sum_scalar(q + i, n - i, thr);
}

// ---- reference bases -> "is N" bits ----
inline uint64_t nmask64_scalar(const uint8_t *r, uint32_t n)
{
    uint64_t m = 0;
    for (uint32_t i = 0; i < n; ++i) m |= (uint64_t)((r[i] | 0x20u) == 'n') << i;
    return m;
}

void nwords_sse2(const uint8_t *r, uint64_t n_words, uint64_t *out)
{
    const __m128i lc = _mm_set1_epi8(0x20), nn = _mm_set1_epi8('n');
    for (uint64_t w = 0; w < n_words; ++w) {
        uint64_t m = 0;
        for (int k = 0; k < 4; ++k) {
            const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(r + 64 * w + 16 * k));
            m |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_or_si128(v, lc), nn)) << (16 * k);
        }
        out[w] = m;
    }
}

__attribute__((target("avx2"))) void nwords_avx2(const uint8_t *r, uint64_t n_words, uint64_t *out)
{
    const __m256i lc = _mm256_set1_epi8(0x20), nn = _mm256_set1_epi8('n');
    for (uint64_t w = 0; w < n_words; ++w) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(r + 64 * w));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(r + 64 * w + 32));
        const uint32_t ma = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_or_si256(a, lc), nn));
        const uint32_t mb = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_or_si256(b, lc), nn));
        out[w] = (uint64_t)ma | ((uint64_t)mb << 32);
    }
}

bool has_avx2()
{
    static const bool v = __builtin_cpu_supports("avx2");
    return v;
}

// one read in one pass: its pass words (bits above n zero) and the sum of its passing bytes
__attribute__((target("avx2"))) uint64_t read_avx2(const uint8_t *q, uint64_t n, uint8_t thr, uint64_t *out)
{
    const __m256i t = _mm256_set1_epi8((char)thr), z = _mm256_setzero_si256();
    __m256i acc = z;
    uint64_t i = 0, w = 0;
    for (; i + 64 <= n; i += 64, ++w) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(q + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(q + i + 32));
        const __m256i ma = _mm256_cmpeq_epi8(_mm256_max_epu8(a, t), a), mb = _mm256_cmpeq_epi8(_mm256_max_epu8(b, t), b);
        out[w] = (uint64_t)(uint32_t)_mm256_movemask_epi8(ma) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(mb) << 32);
        acc = _mm256_add_epi64(acc, _mm256_add_epi64(_mm256_sad_epu8(_mm256_and_si256(a, ma), z), _mm256_sad_epu8(_mm256_and_si256(b, mb), z)));
    }
    uint64_t lanes[4];
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(lanes), acc);
    uint64_t sum = lanes[0] + lanes[1] + lanes[2] + lanes[3];
    if (i < n) {
        uint64_t m = 0;
        uint64_t k = 0;
        if (i + 32 <= n) {
            const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(q + i));
            const __m256i ma = _mm256_cmpeq_epi8(_mm256_max_epu8(a, t), a);
            m = (uint64_t)(uint32_t)_mm256_movemask_epi8(ma);
            const __m256i sd = _mm256_sad_epu8(_mm256_and_si256(a, ma), z);
            _mm256_storeu_si256(reinterpret_cast<__m256i *>(lanes), sd);
            sum += lanes[0] + lanes[1] + lanes[2] + lanes[3];
            k = 32;
        }
        for (; i + k < n; ++k) { const uint8_t v = q[i + k]; if (v >= thr) { m |= 1ull << k; sum += v; } }
        out[w] = m;
    }
    return sum;
}

} // namespace

int qual_pack_level() { return has_avx2() ? 2 : 1; }

void qual_pass_words(const uint8_t *q, uint64_t n_words, uint8_t thr, uint64_t *out, int level)
{
    if (level == 0) { for (uint64_t w = 0; w < n_words; ++w) out[w] = mask64_scalar(q + 64 * w, 64, thr); return; }
    if (level >= 2 && has_avx2()) words_avx2(q, n_words, thr, out); else words_sse2(q, n_words, thr, out);
}

uint64_t qual_pass_partial(const uint8_t *q, uint32_t n, uint8_t thr) { return mask64_scalar(q, n > 64 ? 64 : n, thr); }

uint64_t qual_pass_read(const uint8_t *q, uint64_t n, uint8_t thr, uint64_t *out, int level)
{
    if (level >= 2 && has_avx2()) return read_avx2(q, n, thr, out);
    qual_pass_words(q, n >> 6, thr, out, level);
    if (n & 63ull) out[n >> 6] = qual_pass_partial(q + (n & ~63ull), (uint32_t)(n & 63ull), thr);
    return qual_pass_sum(q, n, thr, level);
}

void ref_n_words(const uint8_t *ref, uint64_t n_bases, uint64_t n_words, uint64_t *out, int level)
{
    const uint64_t whole = std::min<uint64_t>(n_bases >> 6, n_words);
    if (level == 0) { for (uint64_t w = 0; w < whole; ++w) out[w] = nmask64_scalar(ref + 64 * w, 64); }
    else if (level >= 2 && has_avx2()) nwords_avx2(ref, whole, out);
    else nwords_sse2(ref, whole, out);
    for (uint64_t w = whole; w < n_words; ++w) {
        const uint64_t have = n_bases > 64 * w ? std::min<uint64_t>(64, n_bases - 64 * w) : 0;
        uint64_t m = have ? nmask64_scalar(ref + 64 * w, (uint32_t)have) : 0ull;
        if (have < 64) m |= ~0ull << have;                      // beyond the reference: 'N' (mod.rs:79-80)
        out[w] = m;
    }
}

uint64_t qual_pass_sum(const uint8_t *q, uint64_t n, uint8_t thr, int level)
{
    if (level == 0) return sum_scalar(q, n, thr);
    if (level >= 2 && has_avx2()) return sum_avx2(q, n, thr);
    return sum_sse2(q, n, thr);
}

} // namespace dut
