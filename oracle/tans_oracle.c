/* oracle/tans_oracle.c -- TEST INFRASTRUCTURE ONLY (never imported, linked or executed by the product path).
 *
 * Plain-C restatement of the reference's table-ANS coder (cbench/csrc/ans/tans.cpp, an FSE derivative with per-symbol
 * table switching and bypass coding):
 *   count normalisation          tans.cpp:27-147   (Tans_normalizeM2 / Tans_normalizeCount)
 *   encoder tables               tans.cpp:149-226  (Tans_buildCTable)
 *   encoder step / flush         tans.cpp:245-259  (Tans_encodeSymbol, Tans_flushCState)
 *   decoder tables               tans.cpp:261-318  (Tans_buildDTable)
 *   decoder step                 tans.cpp:330-364  (Tans_initDState, Tans_decodeSymbol[Fast])
 *   symbol loop, bypass coding   tans.cpp:527-680 (encode_with_indexes), :722-815 (decode_with_indexes)
 *   bit container                cbench/csrc/FSE/bitstream.h:185-247 (writer), :260-360 (reader)
 * Pinned by tests/golden/tans_kat.npz (bytes produced by the reference's own compiled TansEncoder, oracle/_ref) and by
 * oracle/_ref itself on random inputs (tests/test_oracle_golden.py).
 *
 * The bit stream is handled as what it is -- one little-endian integer: the writer appends fields at the top, the
 * reader removes them from the top (below the end mark).  The reference's 64-bit container with flush / reload is an
 * implementation of exactly that as long as no more than 57 bits are taken between two reloads, which holds for every
 * stream its own encoder can produce (12-bit state + one bypass count + eight 4-bit digits).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TANS_OK 0
#define TANS_ERR_GENERIC -1        /* FSE "GENERIC" (normalisation failed / bad table) */
#define TANS_ERR_DST_SMALL -2      /* FSE "dstSize_tooSmall": capacity n*table_log/8 <= 8 bytes (bitstream.h:192) */
#define TANS_ERR_SRC -3            /* FSE "srcSize_wrong" / missing end mark (bitstream.h:262,270) */
#define TANS_ERR_ARG -4

static unsigned highbit(uint32_t v) { return 31u - (unsigned)__builtin_clz(v); }

/* ---- tans.cpp:27-95: second normalisation method (used when the first over-commits the largest symbol) */
static int normalize_fallback(int16_t *norm, unsigned L, const uint32_t *count, uint64_t total, unsigned nsym)
{
    uint32_t placed = 0;
    uint32_t low_threshold = (uint32_t)(total >> L);
    uint32_t low_one = (uint32_t)((total * 3) >> (L + 1));
    for (unsigned s = 0; s < nsym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= low_threshold) { norm[s] = -1; placed++; total -= count[s]; continue; }
        if (count[s] <= low_one) { norm[s] = 1; placed++; total -= count[s]; continue; }
        norm[s] = -2;
    }
    uint32_t to_place = (1u << L) - placed;
    if (to_place == 0) return TANS_ERR_GENERIC;   /* the reference divides by zero here (SIGFPE) */
    if ((total / to_place) > low_one) {
        low_one = (uint32_t)((total * 3) / (to_place * 2));
        for (unsigned s = 0; s < nsym; s++)
            if (norm[s] == -2 && count[s] <= low_one) { norm[s] = 1; placed++; total -= count[s]; }
        to_place = (1u << L) - placed;
    }
    if (placed == nsym) {   /* everything is rare: the remaining cells go to the most frequent symbol */
        uint32_t best = 0, best_count = 0;
        for (unsigned s = 0; s < nsym; s++)
            if (count[s] > best_count) { best = s; best_count = count[s]; }
        norm[best] = (int16_t)(norm[best] + (int16_t)to_place);
        return TANS_OK;
    }
    {
        const uint64_t vlog = 62 - L;
        const uint64_t mid = (1ull << (vlog - 1)) - 1;
        const uint64_t rstep = (((1ull << vlog) * to_place) + mid) / total;
        uint64_t acc = mid;
        for (unsigned s = 0; s < nsym; s++) {
            if (norm[s] != -2) continue;
            const uint64_t end = acc + (uint64_t)count[s] * rstep;
            const uint32_t weight = (uint32_t)(end >> vlog) - (uint32_t)(acc >> vlog);
            if (weight < 1) return TANS_ERR_GENERIC;
            norm[s] = (int16_t)weight;
            acc = end;
        }
    }
    return TANS_OK;
}

/* ---- tans.cpp:97-147.  Returns TANS_OK, or an error; *rle is set when one symbol holds the whole mass (the reference
 * returns 0 there WITHOUT writing norm, i.e. goes on with an uninitialised table: callers treat it as unsupported). */
int tans_oracle_normalize(const int32_t *freqs, int nsym_i, int table_log, int16_t *norm, int *rle)
{
    static const uint32_t rest_to_beat[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    if (nsym_i < 2 || table_log < 1 || table_log > 15) return TANS_ERR_ARG;
    const unsigned nsym = (unsigned)nsym_i, L = (unsigned)table_log;
    uint32_t *count = (uint32_t *)malloc(sizeof(uint32_t) * nsym);
    uint64_t total = 0;
    for (unsigned s = 0; s < nsym; s++) { count[s] = (uint32_t)freqs[s]; total += count[s]; }
    if (rle) *rle = 0;
    int rc = TANS_OK;
    if (total < 2) { rc = TANS_ERR_ARG; goto done; }
    {   /* FSE_minTableLog, tans.cpp:17-23 */
        const unsigned bits_src = highbit((uint32_t)(total - 1)) + 1, bits_sym = highbit(nsym - 1) + 2;
        if (L < (bits_src < bits_sym ? bits_src : bits_sym)) { rc = TANS_ERR_GENERIC; goto done; }
    }
    {
        const uint64_t scale = 62 - L, step = (1ull << 62) / total, vstep = 1ull << (scale - 20);
        int remaining = 1 << L;
        unsigned largest = 0;
        int16_t largest_p = 0;
        const uint32_t low_threshold = (uint32_t)(total >> L);
        for (unsigned s = 0; s < nsym; s++) {
            if (count[s] == total) { if (rle) *rle = 1; goto done; }
            if (count[s] == 0) { norm[s] = 0; continue; }
            if (count[s] <= low_threshold) { norm[s] = -1; remaining--; continue; }
            int16_t p = (int16_t)(((uint64_t)count[s] * step) >> scale);
            if (p < 8) {
                const uint64_t beat = vstep * rest_to_beat[p];
                p = (int16_t)(p + (((uint64_t)count[s] * step) - ((uint64_t)p << scale) > beat));
            }
            if (p > largest_p) { largest_p = p; largest = s; }
            norm[s] = p;
            remaining -= p;
        }
        if (-remaining >= (norm[largest] >> 1)) rc = normalize_fallback(norm, L, count, total, nsym);
        else norm[largest] = (int16_t)(norm[largest] + (int16_t)remaining);
    }
done:
    free(count);
    return rc;
}

/* one distribution's coding tables */
typedef struct {
    unsigned nsym;
    uint16_t *next_state;      /* [2^L], grouped by symbol: the state reached from sub-range r of symbol s */
    uint32_t *delta_bits;      /* [nsym]  (bits << 16) - first state with that many bits   */
    int32_t *delta_state;      /* [nsym]  start of the symbol's group in next_state - its normalised count */
    uint32_t *d_base;          /* [2^L] decoder: next state before the fresh bits are added */
    uint16_t *d_symbol, *d_bits;
} row_tables;

static void row_free(row_tables *r)
{
    free(r->next_state); free(r->delta_bits); free(r->delta_state); free(r->d_base); free(r->d_symbol); free(r->d_bits);
    memset(r, 0, sizeof(*r));
}

/* the common "spread": cell -> symbol (tans.cpp:166-193 and :280-304 are the same walk) */
static int spread_symbols(const int16_t *norm, unsigned nsym, unsigned L, uint16_t *cell_symbol)
{
    const uint32_t size = 1u << L, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint32_t high = size - 1, pos = 0;
    for (unsigned s = 0; s < nsym; s++)
        if (norm[s] == -1) cell_symbol[high--] = (uint16_t)s;   /* rare symbols own the top cells */
    for (unsigned s = 0; s < nsym; s++)
        for (int k = 0; k < norm[s]; k++) {
            cell_symbol[pos] = (uint16_t)s;
            do pos = (pos + step) & mask; while (pos > high);
        }
    return pos == 0 ? TANS_OK : TANS_ERR_GENERIC;
}

static int row_build(row_tables *r, const int16_t *norm, unsigned nsym, unsigned L)
{
    const uint32_t size = 1u << L;
    memset(r, 0, sizeof(*r));
    r->nsym = nsym;
    r->next_state = (uint16_t *)calloc(size, 2);
    r->delta_bits = (uint32_t *)calloc(nsym, 4);
    r->delta_state = (int32_t *)calloc(nsym, 4);
    r->d_base = (uint32_t *)calloc(size, 4);
    r->d_symbol = (uint16_t *)calloc(size, 2);
    r->d_bits = (uint16_t *)calloc(size, 2);
    uint16_t *cell = (uint16_t *)calloc(size, 2);
    uint32_t *cumul = (uint32_t *)calloc(nsym + 1, 4), *next = (uint32_t *)calloc(nsym, 4);
    int rc = spread_symbols(norm, nsym, L, cell);
    if (rc == TANS_OK) {
        for (unsigned s = 0; s < nsym; s++) {
            const uint32_t c = norm[s] == -1 ? 1u : (uint32_t)(norm[s] > 0 ? norm[s] : 0);
            cumul[s + 1] = cumul[s] + c;
            next[s] = c;
        }
        /* encoder, tans.cpp:195-224 */
        {
            uint32_t *fill = (uint32_t *)malloc(sizeof(uint32_t) * (nsym + 1));
            memcpy(fill, cumul, sizeof(uint32_t) * (nsym + 1));
            for (uint32_t u = 0; u < size; u++) r->next_state[fill[cell[u]]++] = (uint16_t)(size + u);
            free(fill);
        }
        uint32_t total = 0;
        for (unsigned s = 0; s < nsym; s++) {
            if (norm[s] == 0) continue;
            if (norm[s] == -1 || norm[s] == 1) {
                r->delta_bits[s] = (L << 16) - size;
                r->delta_state[s] = (int32_t)total - 1;
                total += 1;
            } else {
                const uint32_t max_bits = L - highbit((uint32_t)norm[s] - 1);
                r->delta_bits[s] = (max_bits << 16) - ((uint32_t)norm[s] << max_bits);
                r->delta_state[s] = (int32_t)total - norm[s];
                total += (uint32_t)norm[s];
            }
        }
        /* decoder, tans.cpp:306-315 */
        for (uint32_t u = 0; u < size; u++) {
            const uint16_t s = cell[u];
            const uint32_t nx = next[s]++;
            const unsigned nb = L - highbit(nx);
            r->d_symbol[u] = s;
            r->d_bits[u] = (uint16_t)nb;
            r->d_base[u] = (nx << nb) - size;
        }
    }
    free(cell); free(cumul); free(next);
    if (rc != TANS_OK) row_free(r);
    return rc;
}

typedef struct {
    int rows, L, bypass, bypass_precision;
    row_tables *row;        /* rows + 1: the last one is the uniform bypass alphabet (tans.cpp:433-459) */
    const int32_t *offsets;
} table_set;

static void set_free(table_set *t)
{
    if (t->row) for (int i = 0; i <= t->rows; i++) row_free(&t->row[i]);
    free(t->row);
    t->row = NULL;
}

static int set_build(table_set *t, const int32_t *freqs, int rows, int stride, const int32_t *nsym, const int32_t *offsets,
                     int L, int bypass, int bypass_precision)
{
    memset(t, 0, sizeof(*t));
    if (rows < 1 || L < 1 || L > 12 || bypass_precision < 1 || bypass_precision > 8) return TANS_ERR_ARG;
    t->rows = rows; t->L = L; t->bypass = bypass; t->bypass_precision = bypass_precision; t->offsets = offsets;
    t->row = (row_tables *)calloc((size_t)rows + 1, sizeof(row_tables));
    int rc = TANS_OK;
    for (int i = 0; i <= rows && rc == TANS_OK; i++) {
        if (i == rows && !bypass) break;
        const int n = i < rows ? nsym[i] : (1 << bypass_precision);
        if (n < 2 || (i < rows && n > stride)) { rc = TANS_ERR_ARG; break; }
        int32_t *f = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
        int16_t *norm = (int16_t *)calloc((size_t)n, sizeof(int16_t));
        for (int s = 0; s < n; s++) f[s] = i < rows ? freqs[(size_t)i * stride + s] : 1;
        int rle = 0;
        rc = tans_oracle_normalize(f, n, L, norm, &rle);
        if (rc == TANS_OK && rle) rc = TANS_ERR_GENERIC;
        if (rc == TANS_OK) rc = row_build(&t->row[i], norm, (unsigned)n, (unsigned)L);
        free(f); free(norm);
    }
    if (rc != TANS_OK) set_free(t);
    return rc;
}

/* AR remap of the table row, ans_interface.hpp:89-104 (same convention as oracle/rans64_oracle.c) */
typedef struct {
    const int32_t *tab; int order, rows, s1;
    const int32_t *ar_indexes, *off[2];
} ar_ctx;

static int32_t ar_lookup(const ar_ctx *ar, int32_t row, const int32_t *sym, int64_t i)
{
    const int32_t a = ar->ar_indexes ? ar->ar_indexes[i] : 0;
    const int32_t v0 = ar->off[0][i] > 0 ? sym[i - ar->off[0][i]] + 1 : 0;
    if (ar->order == 1) return ar->tab[((int64_t)a * ar->rows + row) * ar->s1 + v0];
    const int32_t v1 = ar->off[1][i] > 0 ? sym[i - ar->off[1][i]] + 1 : 0;
    return ar->tab[(((int64_t)a * ar->rows + row) * ar->s1 + v0) * ar->s1 + v1];
}

/* ---- writer: fields are appended above everything written so far */
typedef struct { uint8_t *buf; int64_t cap_bytes; uint64_t nbits; } bit_writer;

static void put_bits(bit_writer *w, uint32_t value, unsigned n)
{
    for (unsigned b = 0; b < n; b++, w->nbits++) {
        const uint64_t byte = w->nbits >> 3;
        if ((int64_t)byte < w->cap_bytes && ((value >> b) & 1u)) w->buf[byte] |= (uint8_t)(1u << (w->nbits & 7));
    }
}

static void encode_step(bit_writer *w, const row_tables *r, uint32_t *state, unsigned symbol)
{   /* tans.cpp:245-252 */
    const uint32_t nb = (*state + r->delta_bits[symbol]) >> 16;
    put_bits(w, *state, nb);
    *state = r->next_state[(int32_t)(*state >> nb) + r->delta_state[symbol]];
}

/* encode_with_indexes (tans.cpp:527-680).  capacity_syms: the symbol count the reference sizes its output buffer with
 * (n for a direct call; the cached symbol count incl. bypass digits for flush(), tans.cpp:686) -- a stream of
 * capacity_syms*L/8 - 8 or more whole bytes is "not storable" and comes back EMPTY (bitstream.h:245). */
int tans_oracle_encode(const int32_t *freqs, int rows, int stride, const int32_t *nsym, const int32_t *offsets, int table_log,
                       int bypass, int bypass_precision, const int32_t *ar_tab, int ar_order, int ar_s1,
                       const int32_t *ar_indexes, const int32_t *ar_off0, const int32_t *ar_off1,
                       const int32_t *symbols, const int32_t *indexes, int64_t n, int64_t capacity_syms,
                       uint8_t *out, int64_t out_cap, int64_t *out_len, int64_t *coded_syms)
{
    table_set T;
    int rc = set_build(&T, freqs, rows, stride, nsym, offsets, table_log, bypass, bypass_precision);
    if (rc) return rc;
    ar_ctx ar = { ar_tab, ar_order, rows, ar_s1, ar_indexes, { ar_off0, ar_off1 } };
    const unsigned L = (unsigned)table_log;
    const uint32_t maxbv = (1u << bypass_precision) - 1u;
    memset(out, 0, (size_t)out_cap);
    bit_writer w = { out, out_cap, 0 };
    uint32_t state = 1u << L;
    int64_t nsyms = 0;
    for (int64_t i = n - 1; i >= 0; i--) {
        int32_t row = indexes[i];
        if (ar_tab) row = ar_lookup(&ar, row, symbols, i);
        const row_tables *r = &T.row[row];
        const int32_t max_value = (int32_t)r->nsym - 1;
        int32_t value = symbols[i] - offsets[row];
        uint32_t raw = 0;
        if (value < 0) { raw = (uint32_t)(-2 * value - 1); value = max_value; }
        else if (value >= max_value) { raw = (uint32_t)(2 * (value - max_value)); value = max_value; }
        if (bypass && value == max_value) {
            /* decode order is: sentinel, digit count (unary in units of maxbv), digits low first -- coded reversed */
            int nb = 0;
            while ((raw >> (nb * bypass_precision)) != 0) ++nb;
            for (int k = nb - 1; k >= 0; k--, nsyms++)
                encode_step(&w, &T.row[rows], &state, (raw >> (k * bypass_precision)) & maxbv);
            encode_step(&w, &T.row[rows], &state, (uint32_t)nb % maxbv);
            nsyms++;
            for (uint32_t k = 0; k < (uint32_t)nb / maxbv; k++, nsyms++) encode_step(&w, &T.row[rows], &state, maxbv);
        }
        encode_step(&w, r, &state, (unsigned)value);
        nsyms++;
    }
    put_bits(&w, state, L);          /* Tans_flushCState, tans.cpp:254-258 */
    put_bits(&w, 1, 1);              /* end mark, bitstream.h:242 */
    if (coded_syms) *coded_syms = nsyms;
    const int64_t cap = (capacity_syms < 0 ? n : capacity_syms) * (int64_t)L / 8;
    if (cap <= 8) rc = TANS_ERR_DST_SMALL;
    else if ((int64_t)(w.nbits >> 3) >= cap - 8) *out_len = 0;
    else {
        *out_len = (int64_t)((w.nbits + 7) >> 3);
        if (*out_len > out_cap) rc = TANS_ERR_ARG;
    }
    set_free(&T);
    return rc;
}

/* ---- reader: fields come off the top, below the end mark */
typedef struct { const uint8_t *buf; int64_t pos; } bit_reader;

static uint32_t take_bits(bit_reader *r, unsigned n)
{
    uint32_t v = 0;
    for (unsigned b = 0; b < n; b++) {
        r->pos--;
        v <<= 1;
        if (r->pos >= 0) v |= (r->buf[r->pos >> 3] >> (r->pos & 7)) & 1u;   /* below the stream: zeros (the reference: undefined) */
    }
    return v;
}

static uint32_t decode_step(bit_reader *br, const row_tables *r, uint32_t *state)
{   /* tans.cpp:338-364 */
    const uint32_t u = *state;
    *state = r->d_base[u] + take_bits(br, r->d_bits[u]);
    return r->d_symbol[u];
}

int tans_oracle_decode(const int32_t *freqs, int rows, int stride, const int32_t *nsym, const int32_t *offsets, int table_log,
                       int bypass, int bypass_precision, const int32_t *ar_tab, int ar_order, int ar_s1,
                       const int32_t *ar_indexes, const int32_t *ar_off0, const int32_t *ar_off1,
                       const uint8_t *stream, int64_t len, const int32_t *indexes, int64_t n, int32_t *out)
{
    if (len < 1 || stream[len - 1] == 0) return TANS_ERR_SRC;
    table_set T;
    int rc = set_build(&T, freqs, rows, stride, nsym, offsets, table_log, bypass, bypass_precision);
    if (rc) return rc;
    ar_ctx ar = { ar_tab, ar_order, rows, ar_s1, ar_indexes, { ar_off0, ar_off1 } };
    bit_reader br = { stream, (len - 1) * 8 + (int64_t)highbit(stream[len - 1]) };
    uint32_t state = take_bits(&br, (unsigned)table_log);
    const uint32_t maxbv = (1u << bypass_precision) - 1u;
    for (int64_t i = 0; i < n; i++) {
        int32_t row = indexes[i];
        if (ar_tab) row = ar_lookup(&ar, row, out, i);
        const row_tables *r = &T.row[row];
        const int32_t max_value = (int32_t)r->nsym - 1;
        int32_t value = (int32_t)decode_step(&br, r, &state);
        if (bypass && value == max_value) {
            uint32_t v = decode_step(&br, &T.row[rows], &state), nb = v;
            while (v == maxbv) { v = decode_step(&br, &T.row[rows], &state); nb += v; }
            uint32_t raw = 0;
            for (uint32_t j = 0; j < nb; j++) raw |= decode_step(&br, &T.row[rows], &state) << (j * bypass_precision);
            value = (int32_t)(raw >> 1);
            if (raw & 1u) value = -value - 1; else value += max_value;
        }
        out[i] = value + offsets[row];
    }
    set_free(&T);
    return TANS_OK;
}

/* table dump for the table-level parity tests: next_state [2^L] u16, delta_bits/delta_state [nsym], decoder triples */
int tans_oracle_tables(const int32_t *freqs, int nsym, int table_log, uint16_t *next_state, uint32_t *delta_bits,
                       int32_t *delta_state, uint32_t *d_base, uint16_t *d_symbol, uint16_t *d_bits)
{
    int16_t *norm = (int16_t *)calloc((size_t)nsym, sizeof(int16_t));
    int rle = 0;
    int rc = tans_oracle_normalize(freqs, nsym, table_log, norm, &rle);
    if (rc == TANS_OK && rle) rc = TANS_ERR_GENERIC;
    row_tables r;
    if (rc == TANS_OK) rc = row_build(&r, norm, (unsigned)nsym, (unsigned)table_log);
    free(norm);
    if (rc) return rc;
    const size_t size = (size_t)1 << table_log;
    memcpy(next_state, r.next_state, size * 2);
    memcpy(delta_bits, r.delta_bits, (size_t)nsym * 4);
    memcpy(delta_state, r.delta_state, (size_t)nsym * 4);
    memcpy(d_base, r.d_base, size * 4);
    memcpy(d_symbol, r.d_symbol, size * 2);
    memcpy(d_bits, r.d_bits, size * 2);
    row_free(&r);
    return TANS_OK;
}
