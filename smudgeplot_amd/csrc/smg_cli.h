/* smg_cli.h -- what the two drop-in executables share: the argv conventions of Myers' ARG_* macros
 * (gene_core.h:32-55), path helpers, table loading with the reference's error messages, the conditioning
 * decision (examine_table + Logex/Symmex plan, PloidyPlot.c:1341-1426 == PloidyList.c:1351-1444) and the
 * engine options.  Header-only (static functions): each executable is one C file plus smg_ktab.c.
 */
#ifndef SMG_CLI_H
#define SMG_CLI_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <stdint.h>

#include "smg_hetmers.h"
#include "smg_ktab.h"

static const char *Prog_Name;          /* set by each main (ARG_INIT, gene_core.h:32-35) */

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

__attribute__((unused)) static void system_x(const char *command)       /* SystemX, gene_core.c:19-24 */
{ if (system(command) != 0)
    { fprintf(stderr, "%s: Command '%s' failed\n", Prog_Name, command);
      exit(1);
    }
}

static int Load_Threads = 1;           /* -T: host threads that read the part files */
static int Load_Lazy = 0;              /* 1: leave the records on disk (smg_ktab_open), the engine pulls them */

static void load_or_die(const char *name, smg_ktab *T)
{ char what[4096];
  switch (Load_Lazy ? smg_ktab_open(name, T, what) : smg_ktab_load_mt(name, T, what, Load_Threads))
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


/* parsed command line common to hetmers / extract_kmer_pairs */
typedef struct smg_cli
{ int   verbose, nthreads, ethresh;
  const char *sort_path;
  char *out;                   /* -o, or NULL */
  int   argc;                  /* positional arguments left in argv[1..argc-1] */
} smg_cli;

__attribute__((unused)) static void smg_cli_parse(int argc, char *argv[], smg_cli *c)
{ int flags[128];
  int i, j, k;
  c->nthreads = 4; c->ethresh = 4; c->sort_path = "/tmp"; c->out = NULL;
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
          c->ethresh = arg_positive(argv[i], "Error-mer threshold");
          break;
        case 'o':
          free(c->out);
          c->out = strdup(argv[i] + 2);
          if (c->out == NULL) exit(1);
          break;
        case 'P':
          c->sort_path = argv[i] + 2;
          break;
        case 'T':
          c->nthreads = arg_positive(argv[i], "Number of threads");
          if (c->nthreads > 64)
            { fprintf(stderr, "%s: Warning, only 64 threads will be used\n", Prog_Name);
              c->nthreads = 64;
            }
          break;
      }
    else
      argv[j++] = argv[i];
  c->argc = j;
  c->verbose = flags['v'];
}

__attribute__((unused)) static void smg_cli_usage_tail(void)
{ fprintf(stderr, "\n");
  fprintf(stderr, "      -o: root name for output table\n");
  fprintf(stderr, "            default is root of <source> argument\n");
  fprintf(stderr, "\n");
  fprintf(stderr, "      -e: count threshold below which k-mers are considered erroneous\n");
  fprintf(stderr, "      -v: verbose mode\n");
  fprintf(stderr, "      -T: number of threads to use\n");
  fprintf(stderr, "      -P: Place all temporary files in directory -P.\n");
  exit(1);
}

/* Open the table, probe it, condition it (on the device by default, with FastK's tools when
   SMUDGEPLOT_USE_FASTK_TOOLS=1), fill the engine options.  Returns the name of a temporary
   conditioned table to Fastrm afterwards (malloc'ed) or NULL.   PloidyPlot.c:1341-1426          */
__attribute__((unused)) static char *smg_cli_open_table(const smg_cli *c, const char *SRC, smg_ktab *T, smg_opts *opts)
{ const char *troot = "";        /* mktemp("._SPAIR.XXXX") yields "" with 4 X's: temps are
                                    literally ".trim"/".symx" in the cwd  */
  int   trim, symm, use_tools = 0, condition = 0;
  char *tname   = (char *) malloc(strlen(SRC) + strlen(troot) + 10);
  char *command = (char *) malloc(strlen(SRC) + strlen(troot) + strlen(c->sort_path) + 100);
  char *input = NULL;

  if (tname == NULL || command == NULL)
    { fprintf(stderr, "%s: Out of memory (Allocating strings)\n", Prog_Name); exit(1); }

  Load_Threads = c->nthreads;
  load_or_die(SRC, T);
  switch (smg_ktab_examine_mt(T, c->ethresh, c->nthreads < 1 ? 1 : c->nthreads > 16 ? 16 : c->nthreads, &trim, &symm)   /* -T is honoured (1..16 probe threads) */)
  { case SMG_KTAB_OK:
      break;
    case SMG_KTAB_NOMEM:
      fprintf(stderr, "%s: Out of memory (Allocating k-mer table)\n", Prog_Name);
      exit(1);
    default:
      fprintf(stderr, "%s: Table file %s is truncated or not a FastK table\n", Prog_Name, SRC);
      exit(1);
  }

  if (c->verbose)
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
    { if (c->verbose)
        { fprintf(stderr, "\n  Trimming k-mers in table with count < %d\n", c->ethresh); fflush(stderr); }
      if (use_tools)
        { sprintf(command, "Logex -T%d '%s.trim=A[%d-]' %s", c->nthreads, troot, c->ethresh, tname);
          system_x(command);
          sprintf(tname, "%s.trim", troot);
        }
      else condition |= SMG_COND_TRIM;
    }
  if (!symm)
    { if (c->verbose)
        { fprintf(stderr, trim ? "\n  Making table symmetric\n" : "\n  Making trimmed table symmetric\n");
          fflush(stderr);
        }
      if (use_tools)
        { sprintf(command, "Symmex -T%d -P%s %s %s.symx", c->nthreads, c->sort_path, tname, troot);
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
    { input = strdup(tname);
      smg_ktab_free(T);
      load_or_die(input, T);
    }
  free(command);
  free(tname);

  memset(opts, 0, sizeof(*opts));
  { const char *g = getenv("SMUDGEPLOT_GPU"), *s = getenv("SMUDGEPLOT_SYMCHECK"), *ng = getenv("SMUDGEPLOT_GPUS");
    opts->device = g ? atoi(g) : 0;
    opts->ngpus = ng ? atoi(ng) : 0;          /* > 1: prefix-shard the table over that many GPUs of the node */
    opts->symcheck = SMG_SYM_HASH;
    if (s && strcasecmp(s, "exact") == 0) opts->symcheck = SMG_SYM_EXACT;
    if (s && strcasecmp(s, "none") == 0) opts->symcheck = SMG_SYM_NONE;
    opts->verbose = c->verbose;
    opts->condition = condition;
    opts->ethresh = c->ethresh;
  }
  return input;
}

/* the table as a source the engine pulls from (records still on disk, or already in memory) */
__attribute__((unused)) static int smg_cli_source_read(void *ctx, int part, int64_t first, int64_t nent, void *dst)
{ return smg_ktab_read((const smg_ktab *) ctx, part, first, nent, dst); }

__attribute__((unused)) static void smg_cli_table_source(const smg_ktab *T, smg_table_source *src, int nthreads)
{ const char *io = getenv("SMUDGEPLOT_IO_THREADS");
  src->kmer = T->kmer; src->ibyte = T->ibyte; src->nparts = T->nparts; src->minval = T->minval;
  src->nels = T->nels;
  src->part_nels = T->part_nels;
  src->prefix_index = T->index;
  src->read = smg_cli_source_read;
  src->ctx = (void *) T;
  /* readers: EIGHT, whatever -T says -- pulling the part files out of the page cache is all the host does for this
     engine; 4 threads (the CLI's default) leave the PCIe link two thirds idle, and more than 8 get in each other's
     way (32 readers: 22.7 instead of 40 GB/s, 0.77 instead of 0.52 s for the 1e9-entry table,
     profiles/r03_e2e_1e9_entries.json).  -T still sets the threads of the conditioning probe; SMUDGEPLOT_IO_THREADS
     overrides the readers.                                                                                      */
  (void) nthreads;
  src->host_threads = io != NULL && atoi(io) > 0 ? atoi(io) : 8;
  if (src->host_threads > 64) src->host_threads = 64;
}

__attribute__((unused)) static void smg_cli_table_view(const smg_ktab *T, smg_table_view *tv)
{ tv->kmer = T->kmer; tv->ibyte = T->ibyte; tv->nparts = T->nparts; tv->minval = T->minval;
  tv->nels = T->nels;
  tv->part_data = (const uint8_t *const *) T->part;
  tv->part_nels = T->part_nels;
  tv->prefix_index = T->index;
}

__attribute__((unused)) static void smg_cli_remove_temp(char *input)
{ if (input != NULL)
    { char *command = (char *) malloc(strlen(input) + 100);
      if (command == NULL) exit(1);
      sprintf(command, "Fastrm %s", input);
      system_x(command);
      free(command);
      free(input);
    }
}

#endif
