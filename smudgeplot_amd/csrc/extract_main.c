/*******************************************************************************************
 *
 *  extract_kmer_pairs -- drop-in replacement for the reference executable of the same name
 *                        (/root/reference/src/lib/PloidyList.c:1207-1583 main), MI355X engine behind it.
 *
 *  `smudgeplot extract` (src/smudgeplot/cli.py:368-382) calls it as
 *      extract_kmer_pairs -o<out> -e<L> -T<t> [-v] [-P<tmp>] <source>[.ktab] <smudges>[.sma]
 *  It runs the same two passes as hetmers, but every pair that would be counted at a pixel annotated
 *  in the .sma file is PRINTED into  <out>.<a>A<b>B.txt  instead (PloidyList.c:424-448, print_het
 *  128-165):  the k-mer of the member with the larger count, "(x/y)" at the variant position.
 *
 *  Host code is plain C; the GPU is reached through smg_hetmers_extract (include/smg_hetmers.h).
 *  No CPU fallback.  Same optional environment as hetmers (smg_cli.h).
 *
 *  The ORDER of the lines inside a file is not defined by the reference (its threads append under a
 *  mutex in schedule order); this program writes them in table order of the pair's lower entry.
 *
 ********************************************************************************************/

#include "smg_cli.h"

static const char *Usage[] = { " [-v] [-T<int(4)>] [-P<dir(/tmp)>]",
                               " [-o<output>] [-e<int(4)>] <source>[.ktab] <smudges>[.sma]" };

typedef struct { int a, b; FILE *f; } Smudge;

int main(int argc, char *argv[])
{ smg_cli c;
  char *OUT, *SRC, *SMA;
  uint16_t *labels;
  Smudge   *smudge;
  int       sm_num = 0, nmax = 100;

  Prog_Name = "extract_kmer_pairs";
  smg_cli_parse(argc, argv, &c);
  if (c.argc != 3)
    { fprintf(stderr, "\nUsage: %s %s\n", Prog_Name, Usage[0]);
      fprintf(stderr, "       %*s %s\n", (int) strlen(Prog_Name), "", Usage[1]);
      smg_cli_usage_tail();
    }
  SRC = argv[1];
  SMA = path_n_root(argv[2], ".sma");
  OUT = c.out != NULL ? c.out : path_n_root(argv[1], ".ktab");

  /* Read in the .sma file and set up the pixel annotations, PloidyList.c:1288-1349 */
  { FILE *f;
    char  buf[1000];
    char *name = (char *) malloc(strlen(SMA) + strlen(OUT) + 64);
    int   i, j, a, b, s;

    labels = (uint16_t *) calloc(SMG_PLOT_CELLS, sizeof(uint16_t));
    smudge = (Smudge *) malloc(sizeof(Smudge) * (size_t) nmax);
    if (labels == NULL || smudge == NULL || name == NULL)
      { fprintf(stderr, "%s: Out of memory (Allocating plot)\n", Prog_Name); exit(1); }

    sprintf(name, "%s.sma", SMA);
    f = fopen(name, "r");
    if (f == NULL)
      { fprintf(stderr, "\n%s: Could not open smudge file %s.sma", Prog_Name, SMA);
        exit(1);
      }
    if (fgets(buf, 1000, f) == NULL) buf[0] = 0;                /* header line */
    while (fgets(buf, 1000, f) != NULL)
      { if (sscanf(buf, " %d %d %*d %dA%dB", &i, &j, &a, &b) != 4)
          { fprintf(stderr, "%s: Cannot parse line '%s'\n", Prog_Name, buf);
            exit(1);
          }
        if (a <= 0 || b <= 0 || a < b)
          { fprintf(stderr, "%s: %dA%dB is not a valid smudge label'\n", Prog_Name, a, b);
            exit(1);
          }
        if (i < 0 || i > SMG_FMAX || j < i || i + j > SMG_SMAX)
          { fprintf(stderr, "%s: (%d,%d) is not a valid pixel coordinate\n", Prog_Name, i, j);
            exit(1);
          }
        for (s = 0; s < sm_num; s++)
          if (smudge[s].a == a && smudge[s].b == b)
            break;
        if (s >= sm_num)
          { if (sm_num >= nmax)
              { nmax += 100;
                smudge = (Smudge *) realloc(smudge, sizeof(Smudge) * (size_t) nmax);
                if (smudge == NULL) exit(1);
              }
            if (sm_num >= 65535)
              { fprintf(stderr, "%s: too many distinct smudge labels\n", Prog_Name); exit(1); }
            smudge[s].a = a;
            smudge[s].b = b;
            sprintf(name, "%s.%dA%dB.txt", OUT, a, b);
            smudge[s].f = fopen(name, "w");
            if (smudge[s].f == NULL)
              { fprintf(stderr, "%s: Cannot open smudge file %s.%dA%dB.txt\n", Prog_Name, OUT, a, b);
                exit(1);
              }
            sm_num += 1;
          }
        labels[(i + j) * SMG_PLOT_COLS + i] = (uint16_t) (s + 1);
      }
    fclose(f);
    free(name);
  }

  { smg_ktab T;
    smg_opts  opts;
    smg_stats stats;
    smg_table_view tv;
    int64_t  *plot;
    uint64_t *rec = NULL;
    int64_t   nrec = 0, r;
    int       rw = 0, k, q;
    char  errbuf[512];
    char *input, *line;
    static const char dna[4] = { 'a', 'c', 'g', 't' };
    int   rc;

    input = smg_cli_open_table(&c, SRC, &T, &opts);
    k = T.kmer;

    if (c.verbose)
      { fprintf(stderr, "\n  Starting to count covariant pairs\n"); fflush(stderr); }

    plot = (int64_t *) malloc(sizeof(int64_t) * SMG_PLOT_CELLS);
    line = (char *) malloc((size_t) k + 16);
    if (plot == NULL || line == NULL)
      { fprintf(stderr, "%s: Out of memory (Allocating plot)\n", Prog_Name); exit(1); }

    smg_cli_table_view(&T, &tv);
    errbuf[0] = 0;
    rc = smg_hetmers_extract(&tv, &opts, labels, plot, &rec, &nrec, &rw, &stats, errbuf, sizeof(errbuf));
    if (rc != SMG_OK)
      { fprintf(stderr, "%s: %s\n", Prog_Name, errbuf[0] ? errbuf : "GPU engine failed");
        exit(1);
      }
    smg_ktab_free(&T);
    smg_cli_remove_temp(input);

    /* print_het, PloidyList.c:128-165: lower-case bases, "(x/y)" at the variant position */
    for (r = 0; r < nrec; r++)
      { const uint64_t *w = rec + (size_t) r * rw;
        const uint64_t  meta = w[rw - 1];
        const int pos = (int) (meta & 0xFF), alt = (int) ((meta >> 8) & 3), lab = (int) (meta >> 16);
        char *o = line;
        for (q = 0; q < k; q++)
          { const int base = (int) ((w[q >> 5] >> (62 - 2 * (q & 31))) & 3);
            if (q == pos)
              { *o++ = '('; *o++ = dna[base]; *o++ = '/'; *o++ = dna[alt]; *o++ = ')'; }
            else
              *o++ = dna[base];
          }
        *o++ = '\n';
        fwrite(line, 1, (size_t) (o - line), smudge[lab - 1].f);
      }
    smg_free(rec);
    free(line);
    free(plot);
  }

  { int s;
    for (s = 0; s < sm_num; s++)
      fclose(smudge[s].f);
  }
  free(smudge);
  free(labels);
  free(SMA);
  free(OUT);
  exit(0);
}
