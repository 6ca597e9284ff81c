/*******************************************************************************************
 *
 *  hetmers_oracle.c  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 *  A plain-C, single-threaded CPU restatement of what the reference `hetmers`
 *  (/root/reference/src/lib/PloidyPlot.c) computes, written from the restated semantics
 *   and NOT from the reference's merge recursion:
 *
 *    table S = {(kmer, count)}                       format F, libfastk.c:786-908, 1230-1269
 *    for every position p in [0,k) group the entries by "k-mer with base p masked out";
 *    inside a group (2..4 entries) every unordered pair (x,y) with cnt_x+cnt_y <= SMAX(1000)
 *    is a one-away pair                              PloidyPlot.c:528-540 (pass 1 rule)
 *    deg(x) = #pairs containing x, kept in a uint8 that wraps mod 256   PloidyPlot.c:163,535
 *    plot[cx+cy][min(cx,cy)] += 1 for pairs with deg(x) <= 1 && deg(y) <= 1
 *                                                    PloidyPlot.c:657-671 / 404-413 (pass 2 rule)
 *    write "<min>\t<sum-min>\t<n>\n" for sum in 0..1000, min in 0..499, n > 0
 *                                                    PloidyPlot.c:1603-1617
 *
 *  It deliberately uses a different algorithm (k independent sorts on the masked key) from
 *  both the reference (4-way merges down a recursion) and the HIP engine (reverse-complement
 *  half-scan over sorted windows), so agreement of the three is meaningful.
 *
 *  Parity pin: this program is checked byte-for-byte against the reference binary compiled
 *  from /root/reference (oracle/Makefile -> oracle/_ref/hetmers_ref) by tests/test_oracle.py
 *  and against the committed fixtures in tests/golden/ produced by that binary.
 *
 *  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may run this.
 *
 *  Usage:  hetmers_oracle [-e<int>] [-x] -o<out> <source>[.ktab]
 *            -x : only print the examine_table decision  "trim=<0|1> symm=<0|1>"
 *                 (PloidyPlot.c:1167-1230) and exit
 *
 ********************************************************************************************/

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <strings.h>

#define SMAX 1000
#define FMAX 500

static int      KMER, KBYTE, TBYTE, IBYTE, NPARTS;
static int64_t  NELS;
static uint8_t *ENT;          /* NELS entries, TBYTE bytes each: KBYTE packed bases + uint16 LE count */

static void die(const char *msg, const char *arg)
{ fprintf(stderr, "hetmers_oracle: %s %s\n", msg, arg ? arg : "");
  exit(1);
}

/* ---- format F loader (restates libfastk.c:786-908 + Current_Entry 1230-1269) ---- */

static void load_table(const char *src)
{ char   *dir, *root, *path;
  const char *slash = strrchr(src, '/');
  size_t  len;
  FILE   *f;
  int32_t hdr[4];
  int64_t *index, ixlen, i, p;

  if (slash) { dir = strndup(src, slash - src); root = strdup(slash + 1); }
  else       { dir = strdup(".");               root = strdup(src); }
  len = strlen(root);
  if (len > 5 && strcasecmp(root + len - 5, ".ktab") == 0)
    root[len - 5] = 0;
  path = malloc(strlen(dir) + strlen(root) + 64);

  sprintf(path, "%s/%s.ktab", dir, root);
  f = fopen(path, "rb");
  if (!f) die("Cannot open k-mer table", src);
  if (fread(hdr, 4, 4, f) != 4) die("short stub", path);
  KMER = hdr[0]; NPARTS = hdr[1]; IBYTE = hdr[3];
  KBYTE = (KMER + 3) >> 2;
  TBYTE = KBYTE + 2;
  ixlen = 1ll << (8 * IBYTE);
  index = malloc(ixlen * 8);
  if ((int64_t) fread(index, 8, ixlen, f) != ixlen) die("short index in", path);
  fclose(f);

  { int pbyte = TBYTE - IBYTE;
    int64_t cap = index[ixlen - 1], n, got = 0;
    uint8_t *rec = malloc(pbyte);
    int64_t pre = 0;

    ENT = malloc((size_t)(cap > 0 ? cap : 1) * TBYTE);
    for (p = 1; p <= NPARTS; p++)
      { int32_t km;
        sprintf(path, "%s/.%s.ktab.%d", dir, root, (int) p);
        f = fopen(path, "rb");
        if (!f) die("Table part is missing:", path);
        if (fread(&km, 4, 1, f) != 1 || fread(&n, 8, 1, f) != 1) die("short part header", path);
        if (km != KMER) die("part k-mer length mismatch", path);
        for (i = 0; i < n; i++)
          { uint8_t *e;
            if (got >= cap) die("more records than the index accounts for in", path);
            if (fread(rec, 1, pbyte, f) != (size_t) pbyte) die("short part", path);
            while (index[pre] <= got) pre++;
            e = ENT + got * TBYTE;
            if (IBYTE == 3) { e[0] = pre >> 16; e[1] = (pre >> 8) & 0xff; e[2] = pre & 0xff; }
            else if (IBYTE == 2) { e[0] = pre >> 8; e[1] = pre & 0xff; }
            else e[0] = pre;
            memcpy(e + IBYTE, rec, pbyte);
            got++;
          }
        fclose(f);
      }
    NELS = got;
    free(rec);
  }
  free(index); free(path); free(dir); free(root);
}

static inline int count_of(int64_t i)
{ const uint8_t *e = ENT + i * TBYTE + KBYTE;
  return e[0] | (e[1] << 8);           /* uint16, little-endian: PloidyPlot.c:529 */
}

/* ---- masked-key sort ---- */

static int MBYTE;          /* byte holding position p  */
static uint8_t MMASK;      /* bits to KEEP in that byte */

static int cmp_masked(const void *a, const void *b)
{ const uint8_t *x = ENT + (*(const int64_t *) a) * TBYTE;
  const uint8_t *y = ENT + (*(const int64_t *) b) * TBYTE;
  int j;
  for (j = 0; j < KBYTE; j++)
    { uint8_t u = x[j], v = y[j];
      if (j == MBYTE) { u &= MMASK; v &= MMASK; }
      if (u != v) return u < v ? -1 : 1;
    }
  return 0;
}

static int same_masked(int64_t a, int64_t b)
{ return cmp_masked(&a, &b) == 0; }

/* ---- reverse complement + binary search, for the examine_table restatement ---- */

static void revcomp(const uint8_t *x, uint8_t *out)
{ int i;
  memset(out, 0, KBYTE);
  for (i = 0; i < KMER; i++)
    { int b = (x[i >> 2] >> (6 - 2 * (i & 3))) & 3;
      int j = KMER - 1 - i;
      out[j >> 2] |= (3 - b) << (6 - 2 * (j & 3));
    }
}

static int64_t find_kmer(const uint8_t *key)
{ int64_t lo = 0, hi = NELS;
  while (lo < hi)
    { int64_t m = (lo + hi) >> 1;
      if (memcmp(ENT + m * TBYTE, key, KBYTE) < 0) lo = m + 1; else hi = m;
    }
  if (lo < NELS && memcmp(ENT + lo * TBYTE, key, KBYTE) == 0) return lo;
  return -1;
}

static void examine(int ethresh, int *trim, int *symm)
{ int64_t frst, last, i, nz = 0;
  static int64_t hist[0x8000];
  uint8_t *rc = malloc(KBYTE);

  if (NELS + 3 < 100000000) { frst = 0; last = NELS; }
  else { frst = NELS / 2 - 50000000; last = NELS / 2 + 50000000; }
  for (i = frst; i < last; i++)
    { int16_t c = (int16_t) count_of(i);          /* read as int16: PloidyPlot.c:1189 */
      if (c >= 0) hist[c]++;
    }
  for (nz = 1; nz < 0x8000 && hist[nz] == 0; nz++) ;
  *trim = (nz >= ethresh);

  /* PloidyPlot.c:1208-1226 walks from entry #1; if the complement is found at another index
     symm=1, if it is absent symm=0.  When entry #1 is its own complement (even k only) the
     reference bumps sidx WITHOUT advancing the stream, so the next iteration finds the same
     k-mer at "another" index and also answers symm=1.  Net effect: symm = rc(entry #1) in T. */
  *symm = 0;
  if (NELS > 1)
    { revcomp(ENT + 1 * TBYTE, rc);
      *symm = (find_kmer(rc) >= 0);
    }
  free(rc);
}

int main(int argc, char **argv)
{ const char *src = NULL, *out = NULL;
  int ethresh = 4, xonly = 0, a;
  int64_t *ord, *pa, *pb, npair = 0, cpair = 1 << 20, i;
  uint8_t *deg;
  static int64_t plot[SMAX + 1][FMAX + 1];
  int p;

  for (a = 1; a < argc; a++)
    if (argv[a][0] == '-')
      switch (argv[a][1])
      { case 'e': ethresh = atoi(argv[a] + 2); break;
        case 'o': out = argv[a] + 2; break;
        case 'x': xonly = 1; break;
        default: break;                     /* -T -P -v accepted and ignored */
      }
    else
      src = argv[a];
  if (!src || (!out && !xonly)) die("usage: hetmers_oracle [-e<int>] [-x] -o<out> <source>[.ktab]", NULL);

  load_table(src);

  if (xonly)
    { int trim, symm;
      examine(ethresh, &trim, &symm);
      printf("trim=%d symm=%d\n", trim, symm);
      return 0;
    }

  ord = malloc(sizeof(int64_t) * (NELS > 0 ? NELS : 1));
  deg = calloc(NELS > 0 ? NELS : 1, 1);
  pa  = malloc(sizeof(int64_t) * cpair);
  pb  = malloc(sizeof(int64_t) * cpair);

  /* pass 1: discover every one-away pair with sum <= SMAX, bump the wrapping uint8 degree */
  for (p = 0; p < KMER; p++)
    { int64_t g0, g1;
      MBYTE = p >> 2;
      MMASK = (uint8_t) ~(3u << (6 - 2 * (p & 3)));
      for (i = 0; i < NELS; i++) ord[i] = i;
      qsort(ord, NELS, sizeof(int64_t), cmp_masked);
      for (g0 = 0; g0 < NELS; g0 = g1)
        { int64_t u, v;
          for (g1 = g0 + 1; g1 < NELS && same_masked(ord[g0], ord[g1]); g1++) ;
          for (u = g0; u < g1; u++)
            for (v = u + 1; v < g1; v++)
              if (count_of(ord[u]) + count_of(ord[v]) <= SMAX)
                { deg[ord[u]] += 1;
                  deg[ord[v]] += 1;
                  if (npair == cpair)
                    { cpair *= 2;
                      pa = realloc(pa, sizeof(int64_t) * cpair);
                      pb = realloc(pb, sizeof(int64_t) * cpair);
                    }
                  pa[npair] = ord[u]; pb[npair] = ord[v]; npair++;
                }
        }
    }

  /* pass 2: histogram the pairs whose two members each have (wrapped) degree <= 1 */
  for (i = 0; i < npair; i++)
    if (deg[pa[i]] <= 1 && deg[pb[i]] <= 1)
      { int cx = count_of(pa[i]), cy = count_of(pb[i]);
        plot[cx + cy][cx < cy ? cx : cy] += 1;
      }

  { char *name = malloc(strlen(out) + 8);
    FILE *f;
    int s, m;
    sprintf(name, "%s.smu", out);
    f = fopen(name, "w");
    if (!f) die("cannot open output", name);
    for (s = 0; s <= SMAX; s++)
      for (m = 0; m < FMAX; m++)             /* i < FMAX: min == 500 is never printed */
        if (plot[s][m] > 0)
          fprintf(f, "%i\t%i\t%lld\n", m, s - m, (long long) plot[s][m]);
    fclose(f);
    free(name);
  }
  return 0;
}
