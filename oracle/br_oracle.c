/*
 * br_oracle.c -- CPU restatement of natir/br's k-mer-spectrum correction path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this file's shared object.  The product
 * path (br_amd/) never links, imports or calls anything under oracle/.
 *
 * Plain C, scalar, single-threaded per call.  Every function cites the reference
 * file:line it follows (paths under /root/reference).  The arithmetic that lives in
 * the un-vendored crates (cocktail @ git f63f0ba, pcon @ git 0184ae7, bio 1.6.0) is
 * restated from their published algorithms; what pins each piece is listed in
 * DESIGN.md ("Oracle pinning").  Pinned: 2-bit codec, parity-canonical, bitset layout,
 * `count > abundance`, One/Two/Graph/GapSize/Greedy unit vectors, the k=11 .solid
 * fixture.  UNPINNED (no reference fixture exists): u8 counter saturate-vs-wrap (we
 * saturate), Greedy positive-fix offsets (rust-bio traceback tie-breaks), FASTA
 * writer line width.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ------------------------------------------------------------------------- */
/* growable byte vector                                                      */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint8_t *p;
    size_t n, cap;
} bvec;

static void bv_init(bvec *v, size_t cap)
{
    v->n = 0;
    v->cap = cap < 16 ? 16 : cap;
    v->p = (uint8_t *)malloc(v->cap);
}
static void bv_push(bvec *v, uint8_t c)
{
    if (v->n == v->cap) {
        v->cap *= 2;
        v->p = (uint8_t *)realloc(v->p, v->cap);
    }
    v->p[v->n++] = c;
}
static void bv_free(bvec *v)
{
    free(v->p);
    v->p = NULL;
    v->n = v->cap = 0;
}

/* ------------------------------------------------------------------------- */
/* u64 membership set (stands in for rustc_hash::FxHashSet<u64>; only        */
/* insert/contains are used by the reference: graph.rs:47, greedy.rs:136,    */
/* gap_size.rs:54)                                                           */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint64_t *slot;
    uint8_t *used;
    size_t cap, n;
} u64set;

static void us_init(u64set *s)
{
    s->cap = 64;
    s->n = 0;
    s->slot = (uint64_t *)calloc(s->cap, sizeof(uint64_t));
    s->used = (uint8_t *)calloc(s->cap, 1);
}
static void us_free(u64set *s)
{
    free(s->slot);
    free(s->used);
}
static size_t us_h(uint64_t x, size_t cap)
{
    x *= 0x9E3779B97F4A7C15ull;
    return (size_t)(x >> 20) & (cap - 1);
}
static int us_contains(const u64set *s, uint64_t x)
{
    size_t h = us_h(x, s->cap);
    while (s->used[h]) {
        if (s->slot[h] == x)
            return 1;
        h = (h + 1) & (s->cap - 1);
    }
    return 0;
}
static void us_insert(u64set *s, uint64_t x);
static void us_grow(u64set *s)
{
    u64set t;
    t.cap = s->cap * 2;
    t.n = 0;
    t.slot = (uint64_t *)calloc(t.cap, sizeof(uint64_t));
    t.used = (uint8_t *)calloc(t.cap, 1);
    for (size_t i = 0; i < s->cap; i++)
        if (s->used[i])
            us_insert(&t, s->slot[i]);
    us_free(s);
    *s = t;
}
static void us_insert(u64set *s, uint64_t x)
{
    if (us_contains(s, x))
        return;
    if ((s->n + 1) * 2 > s->cap)
        us_grow(s);
    size_t h = us_h(x, s->cap);
    while (s->used[h])
        h = (h + 1) & (s->cap - 1);
    s->used[h] = 1;
    s->slot[h] = x;
    s->n++;
}

/* ------------------------------------------------------------------------- */
/* cocktail::kmer (un-vendored, git f63f0ba).  Call sites: correct/mod.rs:61, */
/* 71,80,144; exist/mod.rs:34,58,63; one.rs:67-69.  Conventions pinned by the */
/* k=11 fixture diff and correct/mod.rs:170-181 (A=0, T=2).       */
/* ------------------------------------------------------------------------- */
uint64_t bro_nuc2bit(uint8_t c)
{
    return ((uint64_t)c >> 1) & 3u; /* A=0 C=1 T=2 G=3; any byte folds to 2 bits */
}

uint8_t bro_bit2nuc(uint64_t b)
{
    static const uint8_t tab[4] = {'A', 'C', 'T', 'G'};
    return tab[b & 3u];
}

uint64_t bro_seq2bit(const uint8_t *s, size_t n)
{
    uint64_t kmer = 0;
    for (size_t i = 0; i < n; i++) {
        kmer <<= 2;
        kmer |= bro_nuc2bit(s[i]);
    }
    return kmer;
}

void bro_kmer2seq(uint64_t kmer, int k, uint8_t *out)
{
    for (int i = k - 1; i >= 0; i--) {
        out[i] = bro_bit2nuc(kmer & 3u);
        kmer >>= 2;
    }
}

/* correct/mod.rs:26-42 */
uint64_t bro_mask(int k)
{
    return k <= 0 ? 0 : (k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1));
}

/* correct/mod.rs:110-112 */
static inline uint64_t add_nuc_to_end(uint64_t kmer, uint64_t nuc, int k)
{
    return ((kmer << 2) & bro_mask(k)) ^ nuc;
}

/* cocktail::kmer::revcomp: complement = xor 0b10 per base, then reverse base order */
uint64_t bro_revcomp(uint64_t kmer, int k)
{
    uint64_t x = kmer ^ 0xAAAAAAAAAAAAAAAAull;
    uint64_t r = 0;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (x & 3u);
        x >>= 2;
    }
    return r;
}

/* cocktail::kmer::canonical: the member of {kmer, revcomp} with even popcount
 * (k odd => exactly one has).  Pinned by SURVEY P5.                          */
uint64_t bro_canonical(uint64_t kmer, int k)
{
    if ((__builtin_popcountll(kmer) & 1) == 0)
        return kmer;
    return bro_revcomp(kmer, k);
}

/* pcon index of a forward k-mer: remove_first_bit(canonical) */
uint64_t bro_hash(uint64_t kmer, int k)
{
    return bro_canonical(kmer, k) >> 1;
}

/* ------------------------------------------------------------------------- */
/* pcon::solid::Solid (un-vendored, git 0184ae7): packed canonical bitset,    */
/* bit i of byte i/8 at position i%8 (Lsb0).  set.rs:17-21, set/pcon.rs:188-196 */
/* ------------------------------------------------------------------------- */
typedef struct {
    int k;
    uint64_t nbits;
    uint8_t *bits;
    /* sparse variant for k where 2^(2k-1) bits do not fit a test machine (k = 21: 256 GiB): the same set as a
     * sorted array of canonical hashes; `get` is a binary search.  Checked against the bitset at small k.    */
    uint64_t *sorted;
    size_t n_sorted;
} bro_solid;

uint64_t bro_solid_nbytes(int k)
{
    uint64_t nbits = 1ull << (2 * k - 1);
    return (nbits + 7) / 8;
}

bro_solid *bro_solid_new(int k)
{
    bro_solid *s = (bro_solid *)calloc(1, sizeof(bro_solid));
    s->k = k;
    s->nbits = 1ull << (2 * k - 1);
    s->bits = (uint8_t *)calloc(bro_solid_nbytes(k), 1);
    if (!s->bits) {
        free(s);
        return NULL;
    }
    return s;
}

/* `hashes` = canonical hashes (bro_hash), sorted ascending, unique */
bro_solid *bro_solid_new_sparse(int k, const uint64_t *hashes, size_t n)
{
    bro_solid *s = (bro_solid *)calloc(1, sizeof(bro_solid));
    s->k = k;
    s->nbits = (2 * k - 1 >= 64) ? 0 : (1ull << (2 * k - 1));
    s->sorted = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    if (!s->sorted) {
        free(s);
        return NULL;
    }
    memcpy(s->sorted, hashes, n * sizeof(uint64_t));
    s->n_sorted = n;
    return s;
}

void bro_solid_free(bro_solid *s)
{
    if (!s)
        return;
    free(s->bits);
    free(s->sorted);
    free(s);
}

int bro_solid_k(const bro_solid *s)
{
    return s->k;
}

uint8_t *bro_solid_bits(bro_solid *s)
{
    return s->bits;
}

void bro_solid_set(bro_solid *s, uint64_t kmer, int val)
{
    uint64_t h = bro_hash(kmer, s->k);
    if (val)
        s->bits[h >> 3] |= (uint8_t)(1u << (h & 7));
    else
        s->bits[h >> 3] &= (uint8_t)~(1u << (h & 7));
}

int bro_solid_get(const bro_solid *s, uint64_t kmer)
{
    uint64_t h = bro_hash(kmer, s->k);
    if (s->sorted) {
        size_t lo = 0, hi = s->n_sorted;
        while (lo < hi) {
            size_t mid = lo + (hi - lo) / 2;
            if (s->sorted[mid] < h)
                lo = mid + 1;
            else
                hi = mid;
        }
        return lo < s->n_sorted && s->sorted[lo] == h;
    }
    return (s->bits[h >> 3] >> (h & 7)) & 1;
}

uint64_t bro_solid_popcount(const bro_solid *s)
{
    if (s->sorted)
        return s->n_sorted;
    uint64_t n = 0, nb = bro_solid_nbytes(s->k);
    for (uint64_t i = 0; i < nb; i++)
        n += (uint64_t)__builtin_popcount(s->bits[i]);
    return n;
}

/* set all forward k-mers of a sequence (what the reference unit tests do with
 * cocktail::tokenizer::Tokenizer, e.g. one.rs:96-98)                          */
void bro_solid_set_seq(bro_solid *s, const uint8_t *seq, size_t n)
{
    int k = s->k;
    if (n < (size_t)k)
        return;
    uint64_t kmer = bro_seq2bit(seq, (size_t)k);
    bro_solid_set(s, kmer, 1);
    for (size_t i = (size_t)k; i < n; i++) {
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[i]), k);
        bro_solid_set(s, kmer, 1);
    }
}

/* Solid::from_count: bit i = counts[i] > abundance (main.rs:112-114; `>` pinned by P5) */
bro_solid *bro_solid_from_count(int k, const uint8_t *counts, uint8_t abundance)
{
    bro_solid *s = bro_solid_new(k);
    if (!s)
        return NULL;
    for (uint64_t i = 0; i < s->nbits; i++)
        if (counts[i] > abundance)
            s->bits[i >> 3] |= (uint8_t)(1u << (i & 7));
    return s;
}

/* .solid stream: [k:u8][bits]  (main.rs:117-120, set/pcon.rs:18-25) */
bro_solid *bro_solid_from_bytes(const uint8_t *buf, size_t len)
{
    if (len < 1)
        return NULL;
    int k = buf[0];
    if (k < 1 || k > 31 || len - 1 != bro_solid_nbytes(k))
        return NULL;
    bro_solid *s = bro_solid_new(k);
    if (!s)
        return NULL;
    memcpy(s->bits, buf + 1, len - 1);
    return s;
}

/* wrap an existing bit buffer without copying (used by bench cpu_baseline for 16 GiB sets) */
bro_solid *bro_solid_wrap(int k, uint8_t *bits)
{
    bro_solid *s = (bro_solid *)calloc(1, sizeof(bro_solid));
    s->k = k;
    s->nbits = 1ull << (2 * k - 1);
    s->bits = bits;
    return s;
}
void bro_solid_unwrap(bro_solid *s)
{
    free(s);
}

/* Solid::extend = bitwise OR (set/pcon.rs:88-108) */
void bro_solid_extend(bro_solid *dst, const bro_solid *src)
{
    uint64_t nb = bro_solid_nbytes(dst->k);
    for (uint64_t i = 0; i < nb; i++)
        dst->bits[i] |= src->bits[i];
}

/* ------------------------------------------------------------------------- */
/* pcon::counter::Counter<u8> (un-vendored): counts[hash(canonical)] += 1 per  */
/* k-mer of each record, records shorter than k skipped (main.rs:73-74).       */
/* Saturating at 255 -- saturate-vs-wrap is UNPINNED.              */
/* ------------------------------------------------------------------------- */
uint64_t bro_count_nbytes(int k)
{
    return 1ull << (2 * k - 1);
}

void bro_count_seq(uint8_t *counts, int k, const uint8_t *seq, size_t n)
{
    if (n < (size_t)k)
        return;
    uint64_t kmer = bro_seq2bit(seq, (size_t)k);
    uint64_t h = bro_hash(kmer, k);
    if (counts[h] != 255)
        counts[h]++;
    for (size_t i = (size_t)k; i < n; i++) {
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[i]), k);
        h = bro_hash(kmer, k);
        if (counts[h] != 255)
            counts[h]++;
    }
}

/* canonical hashes of one record, for sort-based builders' tests */
size_t bro_hashes_seq(int k, const uint8_t *seq, size_t n, uint64_t *out)
{
    if (n < (size_t)k)
        return 0;
    uint64_t kmer = bro_seq2bit(seq, (size_t)k);
    size_t m = 0;
    out[m++] = bro_hash(kmer, k);
    for (size_t i = (size_t)k; i < n; i++) {
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[i]), k);
        out[m++] = bro_hash(kmer, k);
    }
    return m;
}

/* presence-only build: Pcon::from_fasta (set/pcon.rs:47-68) */
void bro_solid_set_seq_canonical(bro_solid *s, const uint8_t *seq, size_t n)
{
    bro_solid_set_seq(s, seq, n); /* set() canonicalises, same bits */
}

/* ------------------------------------------------------------------------- */
/* correctors                                                                 */
/* ------------------------------------------------------------------------- */
enum { BRO_ONE = 0, BRO_TWO = 1, BRO_GRAPH = 2, BRO_GREEDY = 3, BRO_GAPSIZE = 4 };

typedef struct {
    uint64_t positions;   /* scan-loop iterations (mod.rs:68)            */
    uint64_t triggers;    /* correct_error invocations (mod.rs:74)       */
    uint64_t fixes;       /* Some(..) returns                            */
    uint64_t fix_i, fix_s, fix_d; /* One: per scenario                   */
    uint64_t rej_alts;    /* alts.len() != 1                             */
    uint64_t rej_noscen;  /* no scenario reaches c                       */
    uint64_t rej_multi;   /* several scenarios, one_more does not isolate */
} bro_stats;

typedef struct {
    const bro_solid *set;
    int method;
    int c;          /* confirm / nb_validate */
    int max_search; /* greedy only */
    bro_stats st;
} bro_corrector;

bro_corrector *bro_corrector_new(const bro_solid *set, int method, int c, int max_search)
{
    bro_corrector *x = (bro_corrector *)calloc(1, sizeof(bro_corrector));
    x->set = set;
    x->method = method;
    x->c = c;
    x->max_search = max_search;
    return x;
}
void bro_corrector_free(bro_corrector *x)
{
    free(x);
}
void bro_corrector_stats(const bro_corrector *x, uint64_t *out9)
{
    out9[0] = x->st.positions;
    out9[1] = x->st.triggers;
    out9[2] = x->st.fixes;
    out9[3] = x->st.fix_i;
    out9[4] = x->st.fix_s;
    out9[5] = x->st.fix_d;
    out9[6] = x->st.rej_alts;
    out9[7] = x->st.rej_noscen;
    out9[8] = x->st.rej_multi;
}

/* result of correct_error: Option<(Vec<u8>, usize)> */
typedef struct {
    int some;
    bvec local;
    size_t offset;
} cerr_t;

/* correct/mod.rs:118-128: solid successors of kmer (k-1 suffix + each of A,C,T,G in code order) */
static int next_nucs(const bro_solid *s, uint64_t kmer, uint64_t alts[4])
{
    int n = 0;
    for (uint64_t a = 0; a < 4; a++)
        if (bro_solid_get(s, add_nuc_to_end(kmer, a, s->k)))
            alts[n++] = a;
    return n;
}

/* correct/mod.rs:114-116 */
static int alt_nucs(const bro_solid *s, uint64_t ori, uint64_t alts[4])
{
    return next_nucs(s, ori >> 2, alts);
}

/* exported for the found_alt_kmer vector (mod.rs:170-181) */
int bro_alt_nucs(const bro_solid *s, uint64_t ori, uint64_t *alts4)
{
    return alt_nucs(s, ori, alts4);
}

/* correct/mod.rs:130-152 */
static size_t error_len(const uint8_t *subseq, size_t len, uint64_t kmer, const bro_solid *s,
                        uint64_t *first_correct)
{
    size_t j = 0;
    for (;;) {
        j += 1;
        if (j >= len)
            break;
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(subseq[j]), s->k);
        if (bro_solid_get(s, kmer))
            break;
    }
    *first_correct = kmer;
    return j;
}

/* ---- Scenario machinery: exist/mod.rs:12-71 -------------------------------- */
/* One: one.rs:33-74.  Two: two.rs:34-328.  Scenario ids follow declaration order
 * (strum EnumIter): One {I,S,D}; Two {II,IS,SS,SD,DD,ICI,ICS,ICD,SCI,SCS,SCD,DCI,DCD}. */
enum { S1_I, S1_S, S1_D, S1_N };
enum { S2_II, S2_IS, S2_SS, S2_SD, S2_DD, S2_ICI, S2_ICS, S2_ICD, S2_SCI, S2_SCS, S2_SCD, S2_DCI, S2_DCD, S2_N };

/* Scenario::apply -> Option<(u64, usize)> */
static int scen_apply(const bro_solid *vs, int two, int sc, uint64_t kmer, const uint8_t *seq, size_t len,
                      uint64_t *okmer, size_t *ooff)
{
    int k = vs->k;
    uint64_t alts[4];
    if (!two) { /* one.rs:57-63 */
        *okmer = kmer;
        *ooff = sc == S1_I ? 2 : (sc == S1_S ? 1 : 0);
        return 1;
    }
    switch (sc) { /* two.rs:89-256 */
    case S2_II:
        *okmer = kmer;
        *ooff = 3;
        return 1;
    case S2_IS:
        *okmer = kmer;
        *ooff = 2;
        return 1;
    case S2_SS:
        if (len < 2)
            return 0;
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[1]), k);
        if (bro_solid_get(vs, kmer))
            return 0;
        if (alt_nucs(vs, kmer, alts) != 1)
            return 0;
        *okmer = add_nuc_to_end(kmer >> 2, alts[0], k);
        *ooff = 2;
        return 1;
    case S2_SD:
        if (len == 0)
            return 0;
        if (alt_nucs(vs, kmer << 2, alts) != 1)
            return 0;
        *okmer = add_nuc_to_end(kmer, alts[0], k);
        *ooff = 1;
        return 1;
    case S2_DD:
        if (alt_nucs(vs, kmer << 2, alts) != 1)
            return 0;
        *okmer = add_nuc_to_end(kmer, alts[0], k);
        *ooff = 0;
        return 1;
    case S2_ICI: {
        if (len < 4)
            return 0;
        uint64_t corr = add_nuc_to_end(kmer, bro_nuc2bit(seq[3]), k);
        if (!bro_solid_get(vs, corr))
            return 0;
        *okmer = corr;
        *ooff = 4;
        return 1;
    }
    case S2_ICS:
        if (len < 4)
            return 0;
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[1]), k);
        if (bro_solid_get(vs, kmer))
            return 0;
        if (alt_nucs(vs, kmer, alts) != 1)
            return 0;
        *okmer = add_nuc_to_end(kmer >> 2, alts[0], k);
        *ooff = 3;
        return 1;
    case S2_ICD: {
        if (len < 4)
            return 0;
        uint64_t second = add_nuc_to_end(kmer, bro_nuc2bit(seq[2]), k);
        if (alt_nucs(vs, second << 2, alts) != 1)
            return 0;
        *okmer = add_nuc_to_end(second, alts[0], k);
        *ooff = 3;
        return 1;
    }
    case S2_SCI:
    case S2_DCI:
        if (len < 4)
            return 0;
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[1]), k);
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[3]), k);
        *okmer = kmer;
        *ooff = 4;
        return 1;
    case S2_SCS:
        if (len < 3)
            return 0;
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[1]), k);
        if (!bro_solid_get(vs, kmer))
            return 0;
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[2]), k);
        if (bro_solid_get(vs, kmer))
            return 0;
        if (alt_nucs(vs, kmer, alts) != 1)
            return 0;
        *okmer = add_nuc_to_end(kmer >> 2, alts[0], k);
        *ooff = 3;
        return 1;
    case S2_SCD:
        if (len < 2)
            return 0;
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[1]), k);
        if (alt_nucs(vs, kmer << 2, alts) != 1)
            return 0;
        *okmer = add_nuc_to_end(kmer, alts[0], k);
        *ooff = 2;
        return 1;
    case S2_DCD:
        if (len < 2)
            return 0;
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[0]), k);
        if (alt_nucs(vs, kmer << 2, alts) != 1)
            return 0;
        *okmer = add_nuc_to_end(kmer, alts[0], k);
        *ooff = 1;
        return 1;
    }
    return 0;
}

/* Scenario::correct -> (Vec<u8>, usize); returns number of bases written to out3 */
static int scen_correct(const bro_solid *vs, int two, int sc, uint64_t kmer, const uint8_t *seq, size_t len,
                        uint8_t out3[3], size_t *ooff)
{
    uint64_t corr = 0;
    size_t off = 0;
    if (!two) { /* one.rs:65-71 */
        out3[0] = bro_bit2nuc(kmer & 3u);
        *ooff = sc == S1_I ? 2 : (sc == S1_S ? 1 : 0);
        return 1;
    }
    switch (sc) { /* two.rs:258-325 */
    case S2_II:
    case S2_IS:
        out3[0] = bro_bit2nuc(kmer & 3u);
        *ooff = 2;
        return 1;
    case S2_SS:
    case S2_SD:
    case S2_DD:
        if (!scen_apply(vs, two, sc, kmer, seq, len, &corr, &off)) {
            fprintf(stderr, "br_oracle: we can't failled her (two.rs:265)\n");
            abort(); /* reference: expect() -> panic=abort */
        }
        out3[0] = bro_bit2nuc((corr & 0xC) >> 2);
        out3[1] = bro_bit2nuc(corr & 3u);
        *ooff = off;
        return 2;
    case S2_ICI:
        out3[0] = bro_bit2nuc(kmer & 3u);
        *ooff = 3;
        return 1;
    case S2_ICD:
        if (!scen_apply(vs, two, sc, kmer, seq, len, &corr, &off))
            abort();
        out3[0] = bro_bit2nuc((corr & 0xC) >> 2);
        out3[1] = bro_bit2nuc(corr & 3u);
        *ooff = off - 1;
        return 2;
    case S2_ICS:
        if (!scen_apply(vs, two, sc, kmer, seq, len, &corr, &off))
            abort();
        out3[0] = bro_bit2nuc((corr & 0xC) >> 2);
        out3[1] = bro_bit2nuc(corr & 3u);
        *ooff = off + 1;
        return 2;
    case S2_SCI:
    case S2_SCS:
    case S2_SCD:
    case S2_DCD:
        if (!scen_apply(vs, two, sc, kmer, seq, len, &corr, &off))
            abort();
        out3[0] = bro_bit2nuc((corr & 0x30) >> 4);
        out3[1] = bro_bit2nuc((corr & 0xC) >> 2);
        out3[2] = bro_bit2nuc(corr & 3u);
        *ooff = off;
        return 3;
    default: /* DCI: two.rs:323 */
        *ooff = 1;
        return 0;
    }
}

/* exist/mod.rs:21-47 */
static size_t scen_score(const bro_solid *vs, int two, int sc, size_t c, uint64_t ori, const uint8_t *seq,
                         size_t len)
{
    uint64_t kmer;
    size_t offset;
    if (!scen_apply(vs, two, sc, ori, seq, len, &kmer, &offset))
        return 0;
    if (!bro_solid_get(vs, kmer))
        return 0;
    if (offset + c > len)
        return 0;
    size_t score = 0;
    for (size_t j = offset; j < offset + c; j++) {
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[j]), vs->k);
        if (bro_solid_get(vs, kmer))
            score++;
        else
            break;
    }
    return score;
}

/* exist/mod.rs:49-70 */
static int scen_one_more(const bro_solid *vs, int two, int sc, size_t c, uint64_t kmer, const uint8_t *seq,
                         size_t len)
{
    uint8_t corr[3];
    size_t offset;
    int nc = scen_correct(vs, two, sc, kmer, seq, len, corr, &offset);
    if (len > c + offset + 1) {
        kmer >>= 2;
        for (int j = 0; j < nc; j++)
            kmer = add_nuc_to_end(kmer, bro_nuc2bit(corr[j]), vs->k);
        for (size_t j = offset; j < offset + c + 1; j++)
            kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[j]), vs->k);
        return bro_solid_get(vs, kmer);
    }
    return 0;
}

/* Exist<S>::correct_error: exist/mod.rs:120-149 */
static void exist_correct_error(bro_corrector *x, int two, uint64_t kmer, const uint8_t *seq, size_t len,
                                cerr_t *r)
{
    const bro_solid *vs = x->set;
    uint64_t alts[4];
    r->some = 0;
    if (alt_nucs(vs, kmer, alts) != 1) {
        x->st.rej_alts++;
        return;
    }
    uint64_t corr = add_nuc_to_end(kmer >> 2, alts[0], vs->k);
    int nscen = two ? S2_N : S1_N;
    int pass[S2_N];
    int np = 0;
    for (int sc = 0; sc < nscen; sc++)
        if (scen_score(vs, two, sc, (size_t)x->c, corr, seq, len) == (size_t)x->c)
            pass[np++] = sc;
    if (np == 0) {
        x->st.rej_noscen++;
        return;
    }
    if (np > 1) {
        int kept[S2_N], nk = 0;
        for (int j = 0; j < np; j++)
            if (scen_one_more(vs, two, pass[j], (size_t)x->c, corr, seq, len))
                kept[nk++] = pass[j];
        if (nk != 1) {
            x->st.rej_multi++;
            return;
        }
        pass[0] = kept[0];
    }
    uint8_t out3[3];
    size_t off;
    int nc = scen_correct(vs, two, pass[0], corr, seq, len, out3, &off);
    r->some = 1;
    r->offset = off;
    for (int j = 0; j < nc; j++)
        bv_push(&r->local, out3[j]);
    if (!two) {
        if (pass[0] == S1_I)
            x->st.fix_i++;
        else if (pass[0] == S1_S)
            x->st.fix_s++;
        else
            x->st.fix_d++;
    }
}

/* Graph::correct_error: graph.rs:44-85 */
static void graph_correct_error(bro_corrector *x, uint64_t kmer, const uint8_t *seq, size_t len, cerr_t *r)
{
    const bro_solid *vs = x->set;
    int k = vs->k;
    uint64_t first_correct, alts[4];
    size_t elen = error_len(seq, len, kmer, vs, &first_correct);
    r->some = 0;

    if (alt_nucs(vs, kmer, alts) != 1) {
        x->st.rej_alts++;
        return;
    }
    u64set viewed;
    us_init(&viewed);
    kmer = add_nuc_to_end(kmer >> 2, alts[0], k);
    bv_push(&r->local, bro_bit2nuc(alts[0]));
    us_insert(&viewed, kmer);

    while (bro_solid_get(vs, kmer)) {
        if (next_nucs(vs, kmer, alts) != 1) {
            us_free(&viewed);
            r->local.n = 0;
            return;
        }
        kmer = add_nuc_to_end(kmer, alts[0], k);
        if (us_contains(&viewed, kmer)) {
            us_free(&viewed);
            r->local.n = 0;
            return;
        }
        us_insert(&viewed, kmer);
        bv_push(&r->local, bro_bit2nuc(alts[0]));
        if (kmer == first_correct)
            break;
    }
    us_free(&viewed);
    r->some = 1;
    r->offset = elen + 1;
}

/* GapSize::ins_sub_correction: gap_size.rs:44-89 */
static void gap_ins_sub(bro_corrector *x, uint64_t kmer, size_t gap_size, cerr_t *r)
{
    const bro_solid *vs = x->set;
    int k = vs->k;
    uint64_t alts[4];
    r->some = 0;
    if (alt_nucs(vs, kmer, alts) != 1) {
        x->st.rej_alts++;
        return;
    }
    uint64_t corr = add_nuc_to_end(kmer >> 2, alts[0], k);
    bv_push(&r->local, bro_bit2nuc(alts[0]));
    u64set viewed;
    us_init(&viewed);
    us_insert(&viewed, corr);
    for (size_t i = 0; i < gap_size; i++) {
        if (next_nucs(vs, corr, alts) != 1) {
            us_free(&viewed);
            r->local.n = 0;
            return;
        }
        corr = add_nuc_to_end(corr, alts[0], k);
        if (us_contains(&viewed, corr)) {
            us_free(&viewed);
            r->local.n = 0;
            return;
        }
        us_insert(&viewed, corr);
        bv_push(&r->local, bro_bit2nuc(alts[0]));
    }
    us_free(&viewed);
    r->some = 1;
    r->offset = r->local.n;
}

/* GapSize::correct_error: gap_size.rs:97-108 */
static void gapsize_correct_error(bro_corrector *x, uint64_t kmer, const uint8_t *seq, size_t len, cerr_t *r)
{
    uint64_t fc;
    size_t elen = error_len(seq, len, kmer, x->set, &fc);
    size_t k = (size_t)x->set->k;
    if (elen < k)
        graph_correct_error(x, kmer, seq, len, r);
    else if (elen == k)
        exist_correct_error(x, 0, kmer, seq, len, r);
    else
        gap_ins_sub(x, kmer, elen - k, r);
}

/* ---- bio 1.6.0 alignment::pairwise::Aligner::global, restated ----------------
 * Affine gaps (open -1, extend -1), match +1 / mismatch -1 (greedy.rs:31-38,63-65),
 * no clipping.  Three layers S/I/D filled column by column over y, `>`-only
 * updates (first maximal candidate wins in the order match/subst, ins, del),
 * traceback from (m,n) through per-layer back pointers.  Ins consumes x, Del
 * consumes y.  Source not in the container: tie-breaks UNPINNED.     */
enum { OP_MATCH = 0, OP_SUBST = 1, OP_DEL = 2, OP_INS = 3 };
enum { TB_START = 0, TB_INS = 1, TB_DEL = 2, TB_SUBST = 3, TB_MATCH = 4, TB_XCLIP = 5 };
#define BIO_MIN_SCORE (-858993459)

typedef struct {
    uint8_t s, i, d;
} tbcell;

/* returns number of ops written (forward order) */
static size_t bio_global(const uint8_t *x, size_t m, const uint8_t *y, size_t n, uint8_t *ops)
{
    const int go = -1, ge = -1;
    size_t W = m + 1;
    tbcell *tb = (tbcell *)calloc((m + 1) * (n + 1), sizeof(tbcell));
    int *S[2], *I[2], *D[2];
    for (int q = 0; q < 2; q++) {
        S[q] = (int *)malloc(W * sizeof(int));
        I[q] = (int *)malloc(W * sizeof(int));
        D[q] = (int *)malloc(W * sizeof(int));
    }
#define TB(i, j) tb[(size_t)(j) * W + (size_t)(i)]
    for (int q = 0; q < 2; q++) {
        for (size_t i = 0; i <= m; i++) {
            S[q][i] = I[q][i] = D[q][i] = BIO_MIN_SCORE;
        }
        S[q][0] = 0;
        if (q == 0) {
            TB(0, 0).s = TB(0, 0).i = TB(0, 0).d = TB_START;
        }
        for (size_t i = 1; i <= m; i++) {
            tbcell c;
            c.s = c.i = c.d = TB_START;
            if (i == 1) {
                I[q][i] = go + ge;
                c.i = TB_START;
            } else {
                int i_score = go + ge * (int)i;
                int c_score = BIO_MIN_SCORE + go + ge;
                if (i_score > c_score) {
                    I[q][i] = i_score;
                    c.i = TB_INS;
                } else {
                    I[q][i] = c_score;
                    c.i = TB_XCLIP;
                }
            }
            if (i == m)
                c.s = TB_XCLIP;
            else
                S[q][i] = BIO_MIN_SCORE;
            if (I[q][i] > S[q][i]) {
                S[q][i] = I[q][i];
                c.s = TB_INS;
            }
            if (q == 0)
                TB(i, 0) = c;
        }
    }
    for (size_t j = 1; j <= n; j++) {
        int cur = (int)(j % 2), prev = 1 - cur;
        {
            tbcell c;
            c.s = c.i = c.d = TB_START;
            I[cur][0] = BIO_MIN_SCORE;
            if (j == 1) {
                D[cur][0] = go + ge;
                c.d = TB_START;
            } else {
                int d_score = go + ge * (int)j;
                int c_score = BIO_MIN_SCORE + go + ge;
                if (d_score > c_score) {
                    D[cur][0] = d_score;
                    c.d = TB_DEL;
                } else {
                    D[cur][0] = c_score;
                    c.d = TB_XCLIP;
                }
            }
            if (D[cur][0] > BIO_MIN_SCORE) {
                S[cur][0] = D[cur][0];
                c.s = TB_DEL;
            } else {
                S[cur][0] = BIO_MIN_SCORE;
                c.s = TB_XCLIP;
            }
            TB(0, j) = c;
        }
        for (size_t i = 1; i <= m; i++)
            S[cur][i] = BIO_MIN_SCORE;
        uint8_t q = y[j - 1];
        for (size_t i = 1; i <= m; i++) {
            uint8_t p = x[i - 1];
            tbcell c;
            c.s = c.i = c.d = TB_START;
            int m_score = S[prev][i - 1] + (p == q ? 1 : -1);

            int i_score = I[cur][i - 1] + ge;
            int s_score = S[cur][i - 1] + go + ge;
            int best_i;
            if (i_score > s_score) {
                best_i = i_score;
                c.i = TB_INS;
            } else {
                best_i = s_score;
                c.i = TB(i - 1, j).s;
            }
            int d_score = D[prev][i] + ge;
            s_score = S[prev][i] + go + ge;
            int best_d;
            if (d_score > s_score) {
                best_d = d_score;
                c.d = TB_DEL;
            } else {
                best_d = s_score;
                c.d = TB(i, j - 1).s;
            }
            c.s = TB_XCLIP;
            int best_s = S[cur][i];
            if (m_score > best_s) {
                best_s = m_score;
                c.s = (p == q) ? TB_MATCH : TB_SUBST;
            }
            if (best_i > best_s) {
                best_s = best_i;
                c.s = TB_INS;
            }
            if (best_d > best_s) {
                best_s = best_d;
                c.s = TB_DEL;
            }
            S[cur][i] = best_s;
            I[cur][i] = best_i;
            D[cur][i] = best_d;
            TB(i, j) = c;
        }
    }
    /* post-pass of the last column (recompute I from possibly changed S): with no
     * clipping S cannot have changed, but the `>` re-check is restated for fidelity */
    {
        size_t j = n;
        int cur = (int)(j % 2);
        for (size_t i = 1; i <= m; i++) {
            int s_score = S[cur][i - 1] + go + ge;
            if (s_score > I[cur][i]) {
                I[cur][i] = s_score;
                TB(i, j).i = TB(i - 1, j).s;
            }
            if (s_score > S[cur][i]) {
                S[cur][i] = s_score;
                TB(i, j).s = TB_INS;
            }
        }
    }
    size_t i = m, j = n, nops = 0;
    uint8_t layer = TB(i, j).s;
    for (;;) {
        uint8_t next;
        if (layer == TB_START)
            break;
        if (layer == TB_INS) {
            ops[nops++] = OP_INS;
            next = TB(i, j).i;
            i--;
        } else if (layer == TB_DEL) {
            ops[nops++] = OP_DEL;
            next = TB(i, j).d;
            j--;
        } else if (layer == TB_MATCH || layer == TB_SUBST) {
            ops[nops++] = (layer == TB_MATCH) ? OP_MATCH : OP_SUBST;
            next = TB(i - 1, j - 1).s;
            i--;
            j--;
        } else {
            break; /* clip states are unreachable with MIN_SCORE clip penalties */
        }
        layer = next;
    }
#undef TB
    for (size_t a = 0, b = nops; a + 1 < b; a++, b--) {
        uint8_t t = ops[a];
        ops[a] = ops[b - 1];
        ops[b - 1] = t;
    }
    for (int q2 = 0; q2 < 2; q2++) {
        free(S[q2]);
        free(I[q2]);
        free(D[q2]);
    }
    free(tb);
    return nops;
}

/* exported for tests */
size_t bro_bio_global(const uint8_t *x, size_t m, const uint8_t *y, size_t n, uint8_t *ops)
{
    return bio_global(x, m, y, n, ops);
}

/* what greedy.rs:66-86 derives from alignment.operations: 1 and *off for Some(off), 0 for None */
static int greedy_outcome(const uint8_t *ops, size_t nops, size_t nb, int64_t *off)
{
    int64_t offset = 0;
    /* operations[before_seq.len()..].windows(2) */
    if (nops >= nb) {
        for (size_t w = nb; w + 1 < nops; w++) {
            uint8_t a = ops[w], b = ops[w + 1];
            if (a == OP_DEL)
                offset -= 1;
            else if (a == OP_INS)
                offset += 1;
            if (a == OP_MATCH && a == b) {
                int64_t offset_corr = 0;
                for (size_t e = nops; e > 0; e--) {
                    uint8_t op = ops[e - 1];
                    if (op == OP_DEL)
                        offset_corr -= 1;
                    else if (op == OP_INS)
                        offset_corr += 1;
                    else
                        break;
                }
                *off = offset - offset_corr;
                return 1;
            }
        }
    }
    return 0;
}

/* ---- audit of the UNPINNED part of Greedy ------------------------------------------------------------------
 * bio_global above restates rust-bio's tie-breaks from memory of its published algorithm; no reference test pins
 * them.  What IS certain is that Aligner::global returns AN optimal alignment under the scoring (match +1, mismatch
 * -1, a gap of length L costs -1 - L).  So: enumerate EVERY optimal operation sequence of a call and derive
 * greedy.rs:66-86's result from each.  If they all agree, the result of that call does not depend on any tie-break
 * -- it is pinned by arithmetic.  The audit counts the calls (and the correct_error invocations) where they differ. */
typedef struct {
    uint64_t calls;          /* match_alignement calls                                          */
    uint64_t multi;          /* ... with more than one optimal operation sequence                */
    uint64_t ambiguous;      /* ... whose optimal sequences give different results               */
    uint64_t capped;         /* ... with more sequences than the enumeration cap (counted ambiguous) */
    uint64_t restated_not_optimal; /* bio_global's own sequence is not among the optimal ones (would be a bug) */
    uint64_t triggers;       /* Greedy correct_error invocations that aligned at least once      */
    uint64_t triggers_amb;   /* ... containing an ambiguous call                                 */
    uint64_t fixes;          /* ... that returned Some                                           */
    uint64_t fixes_amb;      /* ... and contained an ambiguous call                              */
    uint64_t max_sequences;  /* largest number of optimal sequences seen in one call             */
} greedy_audit_t;
static greedy_audit_t g_audit;
static int g_audit_on = 0;
static int g_audit_trigger_amb = 0, g_audit_trigger_calls = 0;

void bro_greedy_audit_enable(int on)
{
    g_audit_on = on;
    if (on)
        memset(&g_audit, 0, sizeof(g_audit));
}
void bro_greedy_audit_get(uint64_t *out10)
{
    memcpy(out10, &g_audit, sizeof(g_audit));
}

#define AUD_MAXDIM 72
#define AUD_CAP 200000
typedef struct {
    const uint8_t *x, *y;
    int m, n;
    size_t nb;
    int S[AUD_MAXDIM][AUD_MAXDIM], I[AUD_MAXDIM][AUD_MAXDIM], D[AUD_MAXDIM][AUD_MAXDIM];
    uint8_t rev[2 * AUD_MAXDIM]; /* operations from the end */
    uint8_t fwd[2 * AUD_MAXDIM];
    uint64_t count;
    int have, some0;   /* first outcome seen */
    int64_t off0;
    int differ;
    const uint8_t *mine; /* bio_global's sequence */
    size_t nmine;
    int mine_found;
} aud_t;
#define AUD_NEG (-100000)

static void aud_leaf(aud_t *a, int nrev)
{
    a->count++;
    for (int q = 0; q < nrev; q++)
        a->fwd[q] = a->rev[nrev - 1 - q];
    int64_t off = 0;
    int some = greedy_outcome(a->fwd, (size_t)nrev, a->nb, &off);
    if (!a->have) {
        a->have = 1;
        a->some0 = some;
        a->off0 = off;
    } else if (some != a->some0 || (some && off != a->off0)) {
        a->differ = 1;
    }
    if ((size_t)nrev == a->nmine && memcmp(a->fwd, a->mine, a->nmine) == 0)
        a->mine_found = 1;
}

/* layer: 0 = S, 1 = I (consumes x), 2 = D (consumes y) */
static void aud_walk(aud_t *a, int i, int j, int layer, int nrev)
{
    if (a->count > AUD_CAP)
        return;
    const int go = -1, ge = -1;
    if (layer == 0) {
        if (i == 0 && j == 0) {
            aud_leaf(a, nrev);
            return;
        }
        if (i > 0 && j > 0) {
            int eq = a->x[i - 1] == a->y[j - 1];
            if (a->S[i][j] == a->S[i - 1][j - 1] + (eq ? 1 : -1)) {
                a->rev[nrev] = eq ? OP_MATCH : OP_SUBST;
                aud_walk(a, i - 1, j - 1, 0, nrev + 1);
            }
        }
        if (i > 0 && a->S[i][j] == a->I[i][j])
            aud_walk(a, i, j, 1, nrev);
        if (j > 0 && a->S[i][j] == a->D[i][j])
            aud_walk(a, i, j, 2, nrev);
    } else if (layer == 1) {
        a->rev[nrev] = OP_INS;
        if (i > 1 && a->I[i][j] == a->I[i - 1][j] + ge)
            aud_walk(a, i - 1, j, 1, nrev + 1);
        if (a->I[i][j] == a->S[i - 1][j] + go + ge) {
            /* leaving the gap: the cell before it must not itself end in the same gap layer, or the same operation
             * sequence would be produced twice with a worse score -- S[i-1][j] is the best over all layers, fine */
            aud_walk(a, i - 1, j, 0, nrev + 1);
        }
    } else {
        a->rev[nrev] = OP_DEL;
        if (j > 1 && a->D[i][j] == a->D[i][j - 1] + ge)
            aud_walk(a, i, j - 1, 2, nrev + 1);
        if (a->D[i][j] == a->S[i][j - 1] + go + ge)
            aud_walk(a, i, j - 1, 0, nrev + 1);
    }
}

/* returns the number of optimal operation sequences (AUD_CAP+1 = more); *differ = their greedy.rs:66-86 results
 * are not all the same; *mine_found = `mine` (an operation sequence, forward order) is one of them */
uint64_t bro_greedy_tie_check(const uint8_t *x, size_t m, const uint8_t *y, size_t n, size_t nb, const uint8_t *mine,
                              size_t nmine, int *differ, int *mine_found)
{
    if (m + 1 > AUD_MAXDIM || n + 1 > AUD_MAXDIM) {
        *differ = 1;
        *mine_found = 0;
        return AUD_CAP + 1;
    }
    aud_t *a = (aud_t *)calloc(1, sizeof(aud_t));
    const int go = -1, ge = -1;
    a->x = x;
    a->y = y;
    a->m = (int)m;
    a->n = (int)n;
    a->nb = nb;
    a->mine = mine;
    a->nmine = nmine;
    for (int i = 0; i <= (int)m; i++)
        for (int j = 0; j <= (int)n; j++) {
            int s = AUD_NEG, ii = AUD_NEG, dd = AUD_NEG;
            if (i == 0 && j == 0)
                s = 0;
            if (i > 0) {
                int e1 = a->I[i - 1][j] + ge, e2 = a->S[i - 1][j] + go + ge;
                ii = e1 > e2 ? e1 : e2;
            }
            if (j > 0) {
                int e1 = a->D[i][j - 1] + ge, e2 = a->S[i][j - 1] + go + ge;
                dd = e1 > e2 ? e1 : e2;
            }
            if (i > 0 && j > 0) {
                int d = a->S[i - 1][j - 1] + (x[i - 1] == y[j - 1] ? 1 : -1);
                if (d > s)
                    s = d;
            }
            if (ii > s)
                s = ii;
            if (dd > s)
                s = dd;
            a->S[i][j] = s;
            a->I[i][j] = ii < AUD_NEG ? AUD_NEG : ii;
            a->D[i][j] = dd < AUD_NEG ? AUD_NEG : dd;
        }
    aud_walk(a, (int)m, (int)n, 0, 0);
    uint64_t cnt = a->count;
    *differ = a->differ || cnt > AUD_CAP;
    *mine_found = a->mine_found;
    free(a);
    return cnt;
}

/* Greedy::match_alignement: greedy.rs:56-89.  returns 1 and *off if Some */
static int greedy_match_alignment(const uint8_t *before, size_t nb, const uint8_t *read, size_t nr,
                                  const uint8_t *corr, size_t nc, int64_t *off)
{
    size_t m = nb + nr, n = nb + nc;
    uint8_t *r = (uint8_t *)malloc(m + 1), *c = (uint8_t *)malloc(n + 1);
    memcpy(r, before, nb);
    memcpy(r + nb, read, nr);
    memcpy(c, before, nb);
    memcpy(c + nb, corr, nc);
    uint8_t *ops = (uint8_t *)malloc(m + n + 2);
    size_t nops = bio_global(r, m, c, n, ops);
    int found = greedy_outcome(ops, nops, nb, off);
    if (g_audit_on) {
        int differ = 0, mine_found = 0;
        uint64_t cnt = bro_greedy_tie_check(r, m, c, n, nb, ops, nops, &differ, &mine_found);
        g_audit.calls++;
        g_audit_trigger_calls++;
        if (cnt > 1)
            g_audit.multi++;
        if (cnt > AUD_CAP)
            g_audit.capped++;
        if (differ) {
            g_audit.ambiguous++;
            g_audit_trigger_amb = 1;
        }
        if (!mine_found && cnt <= AUD_CAP)
            g_audit.restated_not_optimal++;
        if (cnt > g_audit.max_sequences)
            g_audit.max_sequences = cnt;
    }
    free(r);
    free(c);
    free(ops);
    return found;
}

/* Greedy::correct_error: greedy.rs:128-174 */
static void greedy_correct_error(bro_corrector *x, uint64_t kmer, const uint8_t *seq, size_t len, cerr_t *r)
{
    const bro_solid *vs = x->set;
    int k = vs->k;
    uint64_t alts[4];
    r->some = 0;
    if (alt_nucs(vs, kmer, alts) != 1) {
        x->st.rej_alts++;
        return;
    }
    u64set viewed;
    us_init(&viewed);
    uint8_t before[40];
    g_audit_trigger_amb = 0;
    g_audit_trigger_calls = 0;
    bro_kmer2seq(kmer >> 2, k - 1, before);
    kmer = add_nuc_to_end(kmer >> 2, alts[0], k);
    bv_push(&r->local, bro_bit2nuc(alts[0]));
    us_insert(&viewed, kmer);

    for (size_t i = 0; i < (size_t)x->max_search; i++) {
        /* follow_graph: greedy.rs:91-102 */
        if (next_nucs(vs, kmer, alts) == 1) {
            bv_push(&r->local, bro_bit2nuc(alts[0]));
            kmer = add_nuc_to_end(kmer, alts[0], k);
        }
        if (us_contains(&viewed, kmer))
            break;
        us_insert(&viewed, kmer);
        if (len < i)
            break;
        int64_t off;
        if (greedy_match_alignment(before, (size_t)(k - 1), seq, i, r->local.p, r->local.n, &off)) {
            /* check_next_kmers: greedy.rs:104-117 */
            const uint8_t *s2 = seq + i;
            size_t l2 = len - i;
            int ok = 1;
            if (l2 < (size_t)x->c)
                ok = 0;
            else {
                uint64_t km = kmer;
                for (int j = 0; j < x->c; j++) {
                    km = add_nuc_to_end(km, bro_nuc2bit(s2[j]), k);
                    if (!bro_solid_get(vs, km)) {
                        ok = 0;
                        break;
                    }
                }
            }
            if (ok) {
                r->some = 1;
                r->offset = (size_t)((int64_t)r->local.n + off); /* `as usize`: wraps when negative */
                us_free(&viewed);
                if (g_audit_on) {
                    g_audit.triggers++;
                    g_audit.fixes++;
                    g_audit.triggers_amb += (uint64_t)g_audit_trigger_amb;
                    g_audit.fixes_amb += (uint64_t)g_audit_trigger_amb;
                }
                return;
            }
        }
    }
    us_free(&viewed);
    r->local.n = 0;
    if (g_audit_on && g_audit_trigger_calls) {
        g_audit.triggers++;
        g_audit.triggers_amb += (uint64_t)g_audit_trigger_amb;
    }
}

static void correct_error(bro_corrector *x, uint64_t kmer, const uint8_t *seq, size_t len, cerr_t *r)
{
    switch (x->method) {
    case BRO_ONE:
        exist_correct_error(x, 0, kmer, seq, len, r);
        break;
    case BRO_TWO:
        exist_correct_error(x, 1, kmer, seq, len, r);
        break;
    case BRO_GRAPH:
        graph_correct_error(x, kmer, seq, len, r);
        break;
    case BRO_GREEDY:
        greedy_correct_error(x, kmer, seq, len, r);
        break;
    case BRO_GAPSIZE:
        gapsize_correct_error(x, kmer, seq, len, r);
        break;
    default:
        r->some = 0;
    }
}

/* Corrector::correct: correct/mod.rs:53-107.  Returns malloc'd buffer, length in *out_len.
 * Index arithmetic is wrapping, as in the reference's release profile
 * (Cargo.toml:69 overflow-checks=false).  A guard aborts if the scan exceeds a
 * generous iteration budget (the reference would spin).                          */
uint8_t *bro_correct(bro_corrector *x, const uint8_t *seq, size_t len, size_t *out_len)
{
    const bro_solid *vs = x->set;
    size_t k = (size_t)vs->k;
    bvec out;
    bv_init(&out, len + 16);
    if (len < k) {
        for (size_t j = 0; j < len; j++)
            bv_push(&out, seq[j]);
        *out_len = out.n;
        return out.p;
    }
    size_t i = k;
    uint64_t kmer = bro_seq2bit(seq, k);
    for (size_t j = 0; j < k; j++)
        bv_push(&out, seq[j]);
    int previous = bro_solid_get(vs, kmer);
    uint64_t guard = 0, guard_max = 64ull * (uint64_t)len + 4096;
    cerr_t r;
    bv_init(&r.local, 16);
    while (i < len) {
        if (++guard > guard_max) {
            fprintf(stderr, "br_oracle: scan loop exceeded iteration budget\n");
            abort();
        }
        x->st.positions++;
        uint8_t nuc = seq[i];
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(nuc), (int)k);
        if (!bro_solid_get(vs, kmer) && previous) {
            x->st.triggers++;
            r.some = 0;
            r.local.n = 0;
            r.offset = 0;
            correct_error(x, kmer, seq + i, len - i, &r);
            if (r.some) {
                x->st.fixes++;
                kmer >>= 2;
                for (size_t j = 0; j < r.local.n; j++) {
                    kmer = add_nuc_to_end(kmer, bro_nuc2bit(r.local.p[j]), (int)k);
                    bv_push(&out, r.local.p[j]);
                }
                previous = 1;
                i += r.offset;
            } else {
                bv_push(&out, nuc);
                i += 1;
                previous = 0;
            }
        } else {
            previous = bro_solid_get(vs, kmer);
            bv_push(&out, nuc);
            i += 1;
        }
    }
    bv_free(&r.local);
    *out_len = out.n;
    return out.p;
}

void bro_free(void *p)
{
    free(p);
}

/* run_correction's per-record body: lib.rs:42-55 (serial) / 105-117 (rayon).
 * methods chained, then (unless two_side) plain reverse, chain again, reverse back. */
uint8_t *bro_correct_record(bro_corrector **methods, int n_methods, int two_side, const uint8_t *seq,
                            size_t len, size_t *out_len)
{
    uint8_t *cur = (uint8_t *)malloc(len ? len : 1);
    memcpy(cur, seq, len);
    size_t n = len;
    for (int pass = 0; pass < 2; pass++) {
        for (int mi = 0; mi < n_methods; mi++) {
            size_t n2;
            uint8_t *nx = bro_correct(methods[mi], cur, n, &n2);
            free(cur);
            cur = nx;
            n = n2;
        }
        if (two_side)
            break;
        for (size_t a = 0, b = n; a + 1 < b; a++, b--) {
            uint8_t t = cur[a];
            cur[a] = cur[b - 1];
            cur[b - 1] = t;
        }
    }
    *out_len = n;
    return cur;
}

/* batch form used by the parity tests and the CPU baseline:
 * reads concatenated in `bases`, read r = bases[offsets[r] .. offsets[r+1]).
 * Output likewise; out_offsets has n_reads+1 entries.  Returns malloc'd bases.   */
uint8_t *bro_correct_batch(bro_corrector **methods, int n_methods, int two_side, const uint8_t *bases,
                           const uint64_t *offsets, uint32_t n_reads, uint64_t *out_offsets)
{
    bvec out;
    bv_init(&out, (size_t)(offsets[n_reads] - offsets[0]) + 1024);
    out_offsets[0] = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        size_t n;
        uint8_t *c = bro_correct_record(methods, n_methods, two_side, bases + offsets[r],
                                        (size_t)(offsets[r + 1] - offsets[r]), &n);
        if (out.n + n > out.cap) {
            while (out.n + n > out.cap)
                out.cap *= 2;
            out.p = (uint8_t *)realloc(out.p, out.cap);
        }
        memcpy(out.p + out.n, c, n);
        out.n += n;
        free(c);
        out_offsets[r + 1] = out.n;
    }
    return out.p;
}

/* solidity mask of a read: bit j (Lsb0) = solid(k-mer ending at base j+k-1); used to
 * check the GPU scan primitives                                                   */
void bro_solid_mask(const bro_solid *s, const uint8_t *seq, size_t n, uint8_t *mask_bytes)
{
    int k = s->k;
    if (n < (size_t)k)
        return;
    uint64_t kmer = bro_seq2bit(seq, (size_t)k);
    size_t j = 0;
    if (bro_solid_get(s, kmer))
        mask_bytes[j >> 3] |= (uint8_t)(1u << (j & 7));
    for (size_t i = (size_t)k; i < n; i++) {
        kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[i]), k);
        j++;
        if (bro_solid_get(s, kmer))
            mask_bytes[j >> 3] |= (uint8_t)(1u << (j & 7));
    }
}

/* ------------------------------------------------------------------------- */
/* multi-threaded drivers: the CPU baseline of bench.py (never the product).  */
/* The reference's `parallel` feature runs one rayon task per record of an    */
/* 8192-record batch (lib.rs:90-96) and pcon counts into AtomicU8 cells       */
/* (main.rs:73-78); here pthread workers pull blocks of records from an       */
/* atomic cursor.  Results are those of the serial functions above.           */
/* ------------------------------------------------------------------------- */
#include <pthread.h>

typedef struct {
    const bro_solid *set;
    const int *methods;
    int n_methods, c, max_search, two_side;
    const uint8_t *bases;
    const uint64_t *offsets;
    uint32_t n_reads, block;
    uint32_t *cursor;      /* shared */
    uint64_t *out_lens;    /* per read, may be NULL */
    uint64_t out_bases;    /* per thread */
    uint64_t fixes;        /* per thread */
    uint8_t *counts;       /* counting job */
    int k;
    const uint8_t *expect;       /* checking job: somebody else's corrected reads, may be NULL */
    const uint64_t *expect_off;
    uint64_t mismatches;         /* per thread: reads whose bytes differ from `expect` */
    uint32_t first_bad;          /* per thread: the lowest such read (UINT32_MAX: none) */
} mt_job;

static void *mt_correct_worker(void *arg)
{
    mt_job *j = (mt_job *)arg;
    bro_corrector *ms[16];
    for (int m = 0; m < j->n_methods; m++)
        ms[m] = bro_corrector_new(j->set, j->methods[m], j->c, j->max_search);
    for (;;) {
        uint32_t lo = __atomic_fetch_add(j->cursor, j->block, __ATOMIC_RELAXED);
        if (lo >= j->n_reads)
            break;
        uint32_t hi = lo + j->block < j->n_reads ? lo + j->block : j->n_reads;
        for (uint32_t r = lo; r < hi; r++) {
            size_t n;
            uint8_t *cbuf = bro_correct_record(ms, j->n_methods, j->two_side, j->bases + j->offsets[r],
                                               (size_t)(j->offsets[r + 1] - j->offsets[r]), &n);
            if (j->expect) {
                const uint64_t e0 = j->expect_off[r], e1 = j->expect_off[r + 1];
                if (e1 - e0 != (uint64_t)n || (n && memcmp(cbuf, j->expect + e0, n) != 0)) {
                    j->mismatches++;
                    if (r < j->first_bad)
                        j->first_bad = r;
                }
            }
            free(cbuf);
            if (j->out_lens)
                j->out_lens[r] = n;
            j->out_bases += n;
        }
    }
    for (int m = 0; m < j->n_methods; m++) {
        j->fixes += ms[m]->st.fixes;
        bro_corrector_free(ms[m]);
    }
    return NULL;
}

/* corrects every read with the method chain on n_threads threads; returns total corrected bytes,
 * out_lens[r] (optional) = corrected length of read r, *fixes (optional) = Some(..) returns      */
uint64_t bro_correct_batch_mt_check(const bro_solid *set, const int *methods, int n_methods, int c, int max_search,
                                    int two_side, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                                    int n_threads, uint64_t *out_lens, uint64_t *fixes, const uint8_t *expect,
                                    const uint64_t *expect_off, uint64_t *n_mismatch, uint32_t *first_bad);

uint64_t bro_correct_batch_mt(const bro_solid *set, const int *methods, int n_methods, int c, int max_search,
                              int two_side, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                              int n_threads, uint64_t *out_lens, uint64_t *fixes)
{
    return bro_correct_batch_mt_check(set, methods, n_methods, c, max_search, two_side, bases, offsets, n_reads, n_threads,
                                      out_lens, fixes, NULL, NULL, NULL, NULL);
}

/* the same, and every corrected read is compared with somebody else's answer on the way (expect / expect_off, read r at
 * expect[expect_off[r] .. expect_off[r + 1])): *n_mismatch = reads that differ, *first_bad = the lowest of them
 * (UINT32_MAX: none).  This is how a GPU run of 1e5 reads is checked against the oracle read by read without keeping a
 * second gigabyte of corrected bases around (tests/, bench.py's cpu_baseline leg).                                      */
uint64_t bro_correct_batch_mt_check(const bro_solid *set, const int *methods, int n_methods, int c, int max_search,
                                    int two_side, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                                    int n_threads, uint64_t *out_lens, uint64_t *fixes, const uint8_t *expect,
                                    const uint64_t *expect_off, uint64_t *n_mismatch, uint32_t *first_bad)
{
    if (n_threads < 1)
        n_threads = 1;
    if (n_methods > 16)
        n_methods = 16;
    uint32_t cursor = 0;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    mt_job *jobs = (mt_job *)calloc((size_t)n_threads, sizeof(mt_job));
    for (int t = 0; t < n_threads; t++) {
        jobs[t].set = set;
        jobs[t].methods = methods;
        jobs[t].n_methods = n_methods;
        jobs[t].c = c;
        jobs[t].max_search = max_search;
        jobs[t].two_side = two_side;
        jobs[t].bases = bases;
        jobs[t].offsets = offsets;
        jobs[t].n_reads = n_reads;
        jobs[t].block = 16;
        jobs[t].cursor = &cursor;
        jobs[t].out_lens = out_lens;
        jobs[t].expect = expect_off ? expect : NULL;
        jobs[t].expect_off = expect_off;
        jobs[t].first_bad = UINT32_MAX;
        pthread_create(&th[t], NULL, mt_correct_worker, &jobs[t]);
    }
    uint64_t total = 0, fx = 0, bad = 0;
    uint32_t fb = UINT32_MAX;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        total += jobs[t].out_bases;
        fx += jobs[t].fixes;
        bad += jobs[t].mismatches;
        if (jobs[t].first_bad < fb)
            fb = jobs[t].first_bad;
    }
    if (fixes)
        *fixes = fx;
    if (n_mismatch)
        *n_mismatch = bad;
    if (first_bad)
        *first_bad = fb;
    free(jobs);
    free(th);
    return total;
}

static void *mt_count_worker(void *arg)
{
    mt_job *j = (mt_job *)arg;
    const int k = j->k;
    for (;;) {
        uint32_t lo = __atomic_fetch_add(j->cursor, j->block, __ATOMIC_RELAXED);
        if (lo >= j->n_reads)
            break;
        uint32_t hi = lo + j->block < j->n_reads ? lo + j->block : j->n_reads;
        for (uint32_t r = lo; r < hi; r++) {
            const uint8_t *seq = j->bases + j->offsets[r];
            size_t n = (size_t)(j->offsets[r + 1] - j->offsets[r]);
            if (n < (size_t)k)
                continue;
            uint64_t kmer = bro_seq2bit(seq, (size_t)k);
            for (size_t i = (size_t)k;; i++) {
                uint8_t *cell = j->counts + bro_hash(kmer, k);
                uint8_t v = __atomic_load_n(cell, __ATOMIC_RELAXED); /* saturating AtomicU8 increment */
                while (v != 255 && !__atomic_compare_exchange_n(cell, &v, (uint8_t)(v + 1), 1, __ATOMIC_RELAXED,
                                                                __ATOMIC_RELAXED))
                    ;
                if (i >= n)
                    break;
                kmer = add_nuc_to_end(kmer, bro_nuc2bit(seq[i]), k);
            }
        }
    }
    return NULL;
}

/* Counter::count_fasta on n_threads threads into a caller-zeroed 2^(2k-1)-byte table (same table as bro_count_seq) */
void bro_count_batch_mt(uint8_t *counts, int k, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                        int n_threads)
{
    if (n_threads < 1)
        n_threads = 1;
    uint32_t cursor = 0;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    mt_job *jobs = (mt_job *)calloc((size_t)n_threads, sizeof(mt_job));
    for (int t = 0; t < n_threads; t++) {
        jobs[t].counts = counts;
        jobs[t].k = k;
        jobs[t].bases = bases;
        jobs[t].offsets = offsets;
        jobs[t].n_reads = n_reads;
        jobs[t].block = 16;
        jobs[t].cursor = &cursor;
        pthread_create(&th[t], NULL, mt_count_worker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; t++)
        pthread_join(th[t], NULL);
    free(jobs);
    free(th);
}
