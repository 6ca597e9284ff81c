/*******************************************************************************************
 *
 *  smg_condition -- trim and / or symmetrise a FastK k-mer table on the GPU and write the result as a
 *                   FastK table again (format F: stub + part files).
 *
 *  What `hetmers` / `extract_kmer_pairs` of the reference obtain by shelling out to FastK's
 *  Logex '<t>=A[e-]' and Symmex (PloidyPlot.c:1381-1414), as a stand-alone tool: the reference's
 *  executables accept its output as "trimmed and symmetric" and skip their own conditioning.
 *
 *  Usage: smg_condition [-v] [-T<int(4)>] [-e<int(4)>] [-t] [-s] <source>[.ktab] <target>[.ktab]
 *           -e: trim threshold (keep count >= e)          -t: trim only      -s: symmetrise only
 *         (default: both steps; a table that the reference's probe already finds trimmed / symmetric
 *          still goes through the requested steps -- they are idempotent)
 *
 *  The target gets the source's prefix-index width (ibyte) and number of parts.  No CPU fallback.
 *
 ********************************************************************************************/

#include "smg_cli.h"

static int write_all(FILE *f, const void *p, size_t n) { return fwrite(p, 1, n, f) == n ? 0 : -1; }

int main(int argc, char *argv[])
{ int verbose = 0, nthreads = 4, ethresh = 4, only_trim = 0, only_symm = 0;
  int i, j;
  smg_ktab T;
  smg_table_view tv;
  smg_opts opts;
  uint64_t *keys = NULL; uint16_t *cnt = NULL;
  int64_t n = 0;
  int W = 0;
  char errbuf[512];

  Prog_Name = "smg_condition";
  j = 1;
  for (i = 1; i < argc; i++)
    if (argv[i][0] == '-')
      switch (argv[i][1])
      { case 'v': verbose = 1; break;
        case 't': only_trim = 1; break;
        case 's': only_symm = 1; break;
        case 'e': ethresh = arg_positive(argv[i], "Error-mer threshold"); break;
        case 'T': nthreads = arg_positive(argv[i], "Number of threads"); if (nthreads > 64) nthreads = 64; break;
        default:
          fprintf(stderr, "%s: -%c is an illegal option\n", Prog_Name, argv[i][1]);
          exit(1);
      }
    else
      argv[j++] = argv[i];
  if (j != 3 || (only_trim && only_symm))
    { fprintf(stderr, "\nUsage: %s [-v] [-T<int(4)>] [-e<int(4)>] [-t] [-s] <source>[.ktab] <target>[.ktab]\n", Prog_Name);
      fprintf(stderr, "\n      -e: keep the k-mers with count >= e\n      -t: trim only\n      -s: symmetrise only\n");
      exit(1);
    }
  Load_Threads = nthreads;
  load_or_die(argv[1], &T);
  smg_cli_table_view(&T, &tv);
  memset(&opts, 0, sizeof(opts));
  { const char *g = getenv("SMUDGEPLOT_GPU"); opts.device = g ? atoi(g) : 0; }
  opts.ethresh = ethresh;
  opts.condition = (only_symm ? 0 : SMG_COND_TRIM) | (only_trim ? 0 : SMG_COND_SYMM);
  errbuf[0] = 0;
  if (smg_condition_table(&tv, &opts, &keys, &cnt, &n, &W, errbuf, sizeof(errbuf)) != SMG_OK)
    { fprintf(stderr, "%s: %s\n", Prog_Name, errbuf[0] ? errbuf : "GPU engine failed"); exit(1); }
  if (verbose)
    fprintf(stderr, "  %lld -> %lld k-mers (k=%d%s%s)\n", (long long) T.nels, (long long) n, T.kmer,
            only_symm ? "" : ", trimmed", only_trim ? "" : ", symmetrised");

  /* ---- write format F (libfastk.c:786-908 reads it back): stub = kmer, nparts, minval, ibyte, index[];
          part p = kmer, n_p, n_p records of (hbyte suffix bytes + uint16 count) ------------------------- */
  { const int kbyte = T.kbyte, ibyte = T.ibyte, hbyte = kbyte - ibyte, pbyte = hbyte + 2;
    const int nparts = T.nparts > 0 ? T.nparts : 1;
    const int64_t ixlen = T.ixlen;
    int64_t *index = (int64_t *) calloc((size_t) ixlen, sizeof(int64_t));
    int64_t *cut = (int64_t *) malloc(sizeof(int64_t) * (size_t) (nparts + 1));
    uint8_t *rec = (uint8_t *) malloc((size_t) (n > 0 ? n : 1) * pbyte);
    char *root = path_n_root(argv[2], ".ktab");
    const char *slash = strrchr(root, '/');
    char *dir = slash ? strndup(root, (size_t) (slash - root)) : strdup(".");
    const char *base = slash ? slash + 1 : root;
    char *path = (char *) malloc(strlen(root) + 64);
    int64_t e, p;
    int32_t hdr[4];
    FILE *f;
    if (!index || !cut || !rec || !root || !dir || !path)
      { fprintf(stderr, "%s: Out of memory (Allocating k-mer table)\n", Prog_Name); exit(1); }
    for (e = 0; e < n; e++)
      { uint8_t kb[SMG_MAX_KMER / 4 + 8];
        int64_t pre = 0;
        int b;
        for (b = 0; b < kbyte; b++) kb[b] = (uint8_t) (keys[e * W + (b >> 3)] >> (56 - 8 * (b & 7)));
        for (b = 0; b < ibyte; b++) pre = (pre << 8) | kb[b];
        index[pre] += 1;
        memcpy(rec + (size_t) e * pbyte, kb + ibyte, (size_t) hbyte);
        rec[(size_t) e * pbyte + hbyte] = (uint8_t) (cnt[e] & 0xFF);
        rec[(size_t) e * pbyte + hbyte + 1] = (uint8_t) (cnt[e] >> 8);
      }
    for (p = 1; p < ixlen; p++) index[p] += index[p - 1];            /* cumulative END offsets */
    cut[0] = 0; cut[nparts] = n;
    for (p = 1; p < nparts; p++)                                      /* parts break on prefix boundaries */
      { const int64_t target = n / nparts * p;
        int64_t lo = 0, hi = ixlen - 1;
        while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (index[m] < target) lo = m + 1; else hi = m; }
        cut[p] = index[lo] > cut[p - 1] ? index[lo] : cut[p - 1];
      }
    sprintf(path, "%s/%s.ktab", dir, base);
    f = fopen(path, "wb");
    hdr[0] = T.kmer; hdr[1] = nparts; hdr[2] = only_symm ? T.minval : (T.minval > ethresh ? T.minval : ethresh); hdr[3] = ibyte;
    if (f == NULL || write_all(f, hdr, sizeof(hdr)) || write_all(f, index, sizeof(int64_t) * (size_t) ixlen) || fclose(f))
      { fprintf(stderr, "%s: Cannot write %s\n", Prog_Name, path); exit(1); }
    for (p = 0; p < nparts; p++)
      { const int64_t np = cut[p + 1] - cut[p];
        const int32_t km = T.kmer;
        sprintf(path, "%s/.%s.ktab.%d", dir, base, (int) p + 1);
        f = fopen(path, "wb");
        if (f == NULL || write_all(f, &km, 4) || write_all(f, &np, 8)
            || write_all(f, rec + (size_t) cut[p] * pbyte, (size_t) np * pbyte) || fclose(f))
          { fprintf(stderr, "%s: Cannot write %s\n", Prog_Name, path); exit(1); }
      }
    free(index); free(cut); free(rec); free(root); free(dir); free(path);
  }
  smg_free(keys); smg_free(cnt);
  smg_ktab_free(&T);
  exit(0);
}
