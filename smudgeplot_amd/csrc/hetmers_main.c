/*******************************************************************************************
 *
 *  hetmers -- drop-in replacement for the reference `hetmers` executable
 *             (/root/reference/src/lib/PloidyPlot.c:1232-1630 main), MI355X engine behind it.
 *
 *  Same argv grammar, same messages, same exit codes, same `<out>.smu` bytes; the Python CLI
 *  (`smudgeplot hetmers`, src/smudgeplot/cli.py:348-366) calls it unchanged.  Host code is
 *  plain C; the GPU is reached only through the C ABI in include/smg_hetmers.h.
 *
 *  There is NO CPU fallback: without a usable gfx950 device the program prints
 *  "hetmers: <reason>" and exits 1.
 *
 *  Additions that do not touch the reference contract (all optional, environment only, so
 *  the Python CLI stays byte-for-byte unchanged):
 *    SMUDGEPLOT_GPU=<ordinal>            device to use (default 0)
 *    SMUDGEPLOT_USE_FASTK_TOOLS=1        condition with Logex/Symmex/Fastrm like the reference
 *                                        (default: trim + symmetrise on the device)
 *    SMUDGEPLOT_SYMCHECK=hash|exact|none how table symmetry is proven (default hash: 128-bit
 *                                        multiset fingerprint of T against rc(T); exact looks up
 *                                        the complement of every entry; the reference itself only
 *                                        probes entry #1, PloidyPlot.c:1199-1229)
 *    under -v the engine adds one "[smg]" timing line to stderr
 *
 *  Deviations, deliberate:
 *    - the "use it?" prompt stops at EOF on stdin (the reference spins forever there,
 *      PloidyPlot.c:1328);
 *    - "Could not open <out>.smu" names the output (the reference prints its temp root,
 *      PloidyPlot.c:1608).
 *
 ********************************************************************************************/

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <stdint.h>

#include "smg_hetmers.h"
#include "smg_ktab.h"

static const char *Prog_Name = "hetmers";

static const char *Usage[] = { " [-v] [-T<int(4)>] [-P<dir(/tmp)>]",
                               " [-o<output>] [-e<int(4)>] <source>[.ktab]" };

static void usage_and_exit(void)
{ fprintf(stderr, "\nUsage: %s %s\n", Prog_Name, Usage[0]);
  fprintf(stderr, "       %*s %s\n", (int) strlen(Prog_Name), "", Usage[1]);
  fprintf(stderr, "\n");
  fprintf(stderr, "      -o: root name for output table\n");
  fprintf(stderr, "            default is root of <source> argument\n");
  fprintf(stderr, "\n");
  fprintf(stderr, "      -e: count threshold below which k-mers are considered erroneous\n");
  fprintf(stderr, "      -v: verbose mode\n");
  fprintf(stderr, "      -T: number of threads to use\n");
  fprintf(stderr, "      -P: Place all temporary files in directory -P.\n");
  exit(1);
}

/* ARG_POSITIVE, gene_core.h:45-55 */
static int arg_positive(const char *arg, const char *name)
{ char *eptr;
  long  v = strtol(arg + 2, &eptr, 10);
  if (*eptr != '\0' || arg[2] == '\0')
    { fprintf(stderr, "%s: -%c '%s' argument is not an integer\n", Prog_Name, arg[1], arg + 2);
      exit(1);
    }
  if ((int) v <= 0)
    { fprintf(stderr, "%s: %s must be positive (%d)\n", Prog_Name, name, (int) v);
      exit(1);
    }
  return (int) v;
}

/* PathnRoot(name,".ktab"), gene_core.c */
static char *path_n_root(const char *name, const char *suffix)
{ int epos = (int) strlen(name) - (int) strlen(suffix);
  if (epos > 0 && strcasecmp(name + epos, suffix) == 0)
    return strndup(name, (size_t) epos);
  return strdup(name);
}

static void system_x(const char *command)       /* SystemX, gene_core.c:19-24 */
{ if (system(command) != 0)
    { fprintf(stderr, "%s: Command '%s' failed\n", Prog_Name, command);
      exit(1);
    }
}

static void load_or_die(const char *name, smg_ktab *T)
{ char what[4096];
  switch (smg_ktab_load(name, T, what))
  { case SMG_KTAB_OK:
      return;
    case SMG_KTAB_NOSTUB:
      fprintf(stderr, "%s: Cannot open k-mer table %s\n", Prog_Name, name);
      break;
    case SMG_KTAB_NOPART:
      fprintf(stderr, "%s: Table part %s is missing ?\n", Prog_Name, what);
      break;
    case SMG_KTAB_KMISMATCH:
      fprintf(stderr, "%s: Table part %s does not have k-mer length matching stub ?\n", Prog_Name, what);
      break;
    case SMG_KTAB_NOMEM:
      fprintf(stderr, "%s: Out of memory (Allocating k-mer table)\n", Prog_Name);
      break;
    default:
      fprintf(stderr, "%s: Table file %s is truncated or not a FastK table\n", Prog_Name, what);
      break;
  }
  exit(1);
}

int main(int argc, char *argv[])
{ int   VERBOSE, NTHREADS = 4, ETHRESH = 4;
  const char *SORT_PATH = "/tmp";
  char *OUT = NULL, *SRC;
  const char *troot = "";        /* mktemp("._SPAIR.XXXX") yields "" with 4 X's: temps are
                                    literally ".trim"/".symx" in the cwd  */
  int   flags[128];
  int   i, j, k;

  for (i = 0; i < 128; i++) flags[i] = 0;

  j = 1;
  for (i = 1; i < argc; i++)
    if (argv[i][0] == '-')
      switch (argv[i][1])
      { default:                                   /* ARG_FLAGS("vklfs"), gene_core.h:37-43 */
          for (k = 1; argv[i][k] != '\0'; k++)
            { if (strchr("vklfs", argv[i][k]) == NULL)
                { fprintf(stderr, "%s: -%c is an illegal option\n", Prog_Name, argv[i][k]);
                  exit(1);
                }
              flags[(int) argv[i][k]] = 1;
            }
          break;
        case 'e':
          ETHRESH = arg_positive(argv[i], "Error-mer threshold");
          break;
        case 'o':
          free(OUT);
          OUT = strdup(argv[i] + 2);
          if (OUT == NULL) exit(1);
          break;
        case 'P':
          SORT_PATH = argv[i] + 2;
          break;
        case 'T':
          NTHREADS = arg_positive(argv[i], "Number of threads");
          if (NTHREADS > 64)
            { fprintf(stderr, "%s: Warning, only 64 threads will be used\n", Prog_Name);
              NTHREADS = 64;
            }
          break;
      }
    else
      argv[j++] = argv[i];
  argc = j;
  VERBOSE = flags['v'];

  if (argc != 2) usage_and_exit();

  SRC = argv[1];
  if (OUT == NULL) OUT = path_n_root(argv[1], ".ktab");

  /* "If appropriately named het-mer table found then ask if reuse", PloidyPlot.c:1318-1337 */
  { char *name = (char *) malloc(strlen(OUT) + 8);
    FILE *f;
    sprintf(name, "%s.smu", OUT);
    f = fopen(name, "r");
    if (f != NULL)
      { int a, bypass = 0;
        fprintf(stdout, "\n  Found het-table %s.smu, use it? ", OUT);
        fflush(stdout);
        while ((a = getc(stdin)) != '\n' && a != EOF)
          if (a == 'y' || a == 'Y') bypass = 1;
        if (bypass)
          { fprintf(stderr, "\n  Using the found het-table, done\n");
            fclose(f);
            exit(0);
          }
        fclose(f);
      }
    free(name);
  }

  /* Open input table and see if it needs conditioning, PloidyPlot.c:1341-1426 */
  { smg_ktab T;
    int   trim, symm;
    char *tname   = (char *) malloc(strlen(SRC) + strlen(troot) + 10);
    char *command = (char *) malloc(strlen(SRC) + strlen(troot) + strlen(SORT_PATH) + 100);
    char *input = NULL;
    int64_t *plot;
    smg_opts  opts;
    smg_stats stats;
    char errbuf[512];
    int  rc, use_tools = 0, condition = 0;

    if (tname == NULL || command == NULL)
      { fprintf(stderr, "%s: Out of memory (Allocating strings)\n", Prog_Name); exit(1); }

    load_or_die(SRC, &T);
    smg_ktab_examine(&T, ETHRESH, &trim, &symm);

    if (VERBOSE)
      { fprintf(stderr, "\n  The input table is");
        if (trim)
          fprintf(stderr, symm ? " trimmed and symmetric\n" : " trimmed but not symmetric\n");
        else
          fprintf(stderr, symm ? " untrimmed yet symmetric\n" : " untrimmed and not symmetric\n");
      }

    sprintf(tname, "%s", SRC);

    /* Conditioning.  The reference delegates it to FastK's Logex / Symmex / Fastrm through system(3)
       (PloidyPlot.c:1381-1414); those tools are not part of smudgeplot.  Here the same two steps run on
       the device right after the table is decoded (smg_opts.condition), with the same progress lines
       and no temporary tables.  SMUDGEPLOT_USE_FASTK_TOOLS=1 restores the reference's shell-outs
       (identical command strings, temp tables ".trim" / ".symx" in the cwd).                      */
    { const char *ft = getenv("SMUDGEPLOT_USE_FASTK_TOOLS");
      use_tools = ft != NULL && atoi(ft) != 0;
    }
    if (!trim)
      { if (VERBOSE)
          { fprintf(stderr, "\n  Trimming k-mers in table with count < %d\n", ETHRESH); fflush(stderr); }
        if (use_tools)
          { sprintf(command, "Logex -T%d '%s.trim=A[%d-]' %s", NTHREADS, troot, ETHRESH, tname);
            system_x(command);
            sprintf(tname, "%s.trim", troot);
          }
        else condition |= SMG_COND_TRIM;
      }
    if (!symm)
      { if (VERBOSE)
          { fprintf(stderr, trim ? "\n  Making table symmetric\n" : "\n  Making trimmed table symmetric\n");
            fflush(stderr);
          }
        if (use_tools)
          { sprintf(command, "Symmex -T%d -P%s %s %s.symx", NTHREADS, SORT_PATH, tname, troot);
            system_x(command);
            if (!trim)
              { sprintf(command, "Fastrm %s.trim", troot);
                system_x(command);
              }
            sprintf(tname, "%s.symx", troot);
          }
        else condition |= SMG_COND_SYMM;
      }
    if (use_tools && !(symm && trim))
      { input = tname;
        smg_ktab_free(&T);
        load_or_die(input, &T);
      }

    if (VERBOSE)
      { fprintf(stderr, "\n  Starting to count covariant pairs\n"); fflush(stderr); }

    plot = (int64_t *) malloc(sizeof(int64_t) * SMG_PLOT_CELLS);
    if (plot == NULL)
      { fprintf(stderr, "%s: Out of memory (Allocating plot)\n", Prog_Name); exit(1); }

    memset(&opts, 0, sizeof(opts));
    { const char *g = getenv("SMUDGEPLOT_GPU"), *s = getenv("SMUDGEPLOT_SYMCHECK");
      opts.device = g ? atoi(g) : 0;
      opts.symcheck = SMG_SYM_HASH;
      if (s && strcasecmp(s, "exact") == 0) opts.symcheck = SMG_SYM_EXACT;
      if (s && strcasecmp(s, "none") == 0) opts.symcheck = SMG_SYM_NONE;
      opts.verbose = VERBOSE;
      opts.condition = condition;
      opts.ethresh = ETHRESH;
    }
    { smg_table_view tv;
      tv.kmer = T.kmer; tv.ibyte = T.ibyte; tv.nparts = T.nparts; tv.minval = T.minval;
      tv.nels = T.nels;
      tv.part_data = (const uint8_t *const *) T.part;
      tv.part_nels = T.part_nels;
      tv.prefix_index = T.index;
      errbuf[0] = 0;
      rc = smg_hetmers_run(&tv, &opts, plot, &stats, errbuf, sizeof(errbuf));
    }
    if (rc != SMG_OK)
      { fprintf(stderr, "%s: %s\n", Prog_Name, errbuf[0] ? errbuf : "GPU engine failed");
        exit(1);
      }
    smg_ktab_free(&T);

    if (input != NULL)
      { sprintf(command, "Fastrm %s", input);
        system_x(command);
      }
    free(command);
    free(tname);

    if (VERBOSE)
      { fprintf(stderr, "\n  Count complete, outputting table\n"); fflush(stderr); }

    /* writer, PloidyPlot.c:1603-1617: sum ascending, min ascending, min == 500 never printed */
    { char *name = (char *) malloc(strlen(OUT) + 8);
      FILE *f;
      int   a;
      sprintf(name, "%s.smu", OUT);
      f = fopen(name, "w");
      if (f == NULL)
        { fprintf(stderr, "Could not open %s.smu\n", OUT);
          exit(1);
        }
      for (a = 0; a <= SMG_SMAX; a++)
        for (i = 0; i < SMG_FMAX; i++)
          if (plot[a * SMG_PLOT_COLS + i] > 0)
            fprintf(f, "%i\t%i\t%lld\n", i, a - i, (long long) plot[a * SMG_PLOT_COLS + i]);
      fclose(f);
      free(name);
    }
    free(plot);
  }

  free(OUT);
  exit(0);
}
