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
 *    SMUDGEPLOT_IO_THREADS=<n>           threads that read the part files (default: the -T value, at least 8)
 *    SMUDGEPLOT_GPUS=<n>                 prefix-shard the table over n GPUs of the node
 *    SMUDGEPLOT_ONE_PROCESS=1            everything in the one process the caller started (see below)
 *    under -v the engine adds one "[smg]" timing line to stderr
 *
 *  Two processes (round 5).  The HIP runtime takes 80-240 ms to start and ~75 ms to take down, whatever the program does
 *  with it, and neither overlaps with anything inside ONE process: started in a thread beside the table probe the two get
 *  in each other's way (they share an address space; profiles/r04_hip_startup.txt).  So the program the caller started
 *  forks a WORKER first thing -- which starts the runtime at once -- and itself opens the table (stub, the 134 MB prefix
 *  index read straight into a mapping both share, part headers) and runs the reference's conditioning probe, with the
 *  reference's messages.  Then the worker opens the parts for itself, streams them into HBM, runs the passes, writes the
 *  .smu and reports its exit status through a pipe; the starter leaves with that status at once, while the worker is
 *  still handing its device context back.  When `hetmers` returns, the .smu is complete and closed, exactly as before;
 *  what is still going on for a few tens of milliseconds is the release of the GPU by a process nobody waits for.
 *  Any failure to set this up (fork, mmap, pipe) falls back to the one process.
 *
 *  The caller still deals with ONE process (round 6).  The reference is a single process without signal handlers
 *  (PloidyPlot.c:1232-1630; cli.py:57-72 waits for it with subprocess.run): killed, it is gone and nothing writes
 *  <out>.smu afterwards.  So the worker asks the kernel for SIGTERM when the starter dies (PR_SET_PDEATHSIG, and looks at
 *  getppid() once more behind it), the starter hands SIGINT / SIGTERM / SIGHUP / SIGQUIT on to the worker and then dies
 *  of the signal itself, a worker that is signalled while <out>.smu is open unlinks the half-written file, and a worker
 *  that dies without a word costs the starter exit status 1 and the half-written file.  The death signal is withdrawn
 *  once the .smu is closed: from there on the worker only hands the device back.
 *
 *  Deviations, deliberate:
 *    - the "use it?" prompt stops at EOF on stdin (the reference spins forever there,
 *      PloidyPlot.c:1328);
 *    - "Could not open <out>.smu" names the output (the reference prints its temp root,
 *      PloidyPlot.c:1608).
 *
 ********************************************************************************************/

#include <time.h>
#include <unistd.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/mman.h>
#include <sys/types.h>
#include <sys/wait.h>
#include "smg_cli.h"

static double now_s(void)
{ struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

static double real_s(void)
{ struct timespec t;
  clock_gettime(CLOCK_REALTIME, &t);
  return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

/* what the starter hands to the worker (a mapping both share; the index is the only big thing in it) */
#define SHARED_IXWORDS (1ll << 24)
typedef struct
{ smg_opts opts;
  int      have_input;
  volatile int smu_state;        /* 0 not opened yet, 1 open and being written, 2 complete and closed */
  double   t_probe;
  char     input[4096], name[4096];
  int64_t  index[SHARED_IXWORDS];
} Shared;

/* run the engine on the open table T (conditioned, or to be conditioned on the device: opts), write <OUT>.smu: the part of
   main() behind the table probe.  Returns the exit status. */
/* the output while it is open: a signalled process takes the half-written file with it (handlers below) */
static char                  Smu_Path[4200];
static volatile sig_atomic_t Smu_Open = 0;
static volatile int         *Smu_State = NULL;      /* the starter's view of the same (two processes), or NULL */

static void smu_mark(int state)
{ Smu_Open = state == 1;
  if (Smu_State != NULL) *Smu_State = state;
}

static int run(const smg_cli *c, const char *OUT, smg_ktab *Tp, const smg_opts *opts, char *input,
               double t_start, double rt_start, double t_probe)
{ smg_ktab T = *Tp;
  smg_stats stats;
  smg_table_source src;
  int64_t *plot;
  char  errbuf[512];
  int   rc, i;
  double t_engine;

  if (c->verbose)
    { fprintf(stderr, "\n  Starting to count covariant pairs\n"); fflush(stderr); }

  plot = (int64_t *) malloc(sizeof(int64_t) * SMG_PLOT_CELLS);
  if (plot == NULL)
    { fprintf(stderr, "%s: Out of memory (Allocating plot)\n", Prog_Name); return 1; }

  smg_cli_table_source(&T, &src, c->nthreads);
  errbuf[0] = 0;
  rc = smg_hetmers_run_source(&src, opts, plot, &stats, errbuf, sizeof(errbuf));
  if (rc != SMG_OK)
    { fprintf(stderr, "%s: %s\n", Prog_Name, errbuf[0] ? errbuf : "GPU engine failed");
      return 1;
    }
  t_engine = now_s();
  smg_ktab_free(&T);
  smg_cli_remove_temp(input);

  if (c->verbose)
    { fprintf(stderr, "\n  Count complete, outputting table\n"); fflush(stderr); }

  /* writer, PloidyPlot.c:1603-1617: sum ascending, min ascending, min == 500 never printed */
  { char *fname = (char *) malloc(strlen(OUT) + 8);
    FILE *f;
    int   a;
    sprintf(fname, "%s.smu", OUT);
    snprintf(Smu_Path, sizeof(Smu_Path), "%s", strlen(fname) < sizeof(Smu_Path) ? fname : "");
    f = fopen(fname, "w");
    if (f == NULL)
      { fprintf(stderr, "Could not open %s.smu\n", OUT);
        return 1;
      }
    smu_mark(1);
    for (a = 0; a <= SMG_SMAX; a++)
      for (i = 0; i < SMG_FMAX; i++)
        if (plot[a * SMG_PLOT_COLS + i] > 0)
          fprintf(f, "%i\t%i\t%lld\n", i, a - i, (long long) plot[a * SMG_PLOT_COLS + i]);
    fclose(f);
    smu_mark(2);
    free(fname);
  }
  free(plot);
  if (c->verbose)               /* where the wall time of the process went, next to the engine's own lines */
    fprintf(stderr, "  [smg] process %.1f ms since main(): arguments + stub, index and conditioning probe %.1f, engine call %.1f, "
            ".smu writer %.1f; main() entered at %.3f, left at %.3f (CLOCK_REALTIME: what lies outside is the loader and exit())\n",
            (now_s() - t_start) * 1e3, (t_probe - t_start) * 1e3, (t_engine - t_probe) * 1e3,
            (now_s() - t_engine) * 1e3, rt_start, real_s());
  return 0;
}

/* ---- one process as far as the caller can tell (two processes) ---- */

static volatile pid_t Worker_Pid = -1;
static const int Handed_On[] = { SIGINT, SIGTERM, SIGHUP, SIGQUIT };

/* worker: take a half-written .smu along and go (unlink and _exit are async-signal-safe) */
static void worker_signalled(int sig)
{ if (Smu_Open && Smu_Path[0] != 0) unlink(Smu_Path);
  _exit(128 + sig);
}

/* starter: hand the signal on, then die of it as the reference's one process would */
static void starter_signalled(int sig)
{ if (Worker_Pid > 0) kill(Worker_Pid, sig);
  signal(sig, SIG_DFL);
  raise(sig);
}

static void install(void (*handler)(int))
{ struct sigaction sa;
  size_t i;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = handler;
  sigemptyset(&sa.sa_mask);
  for (i = 0; i < sizeof(Handed_On) / sizeof(Handed_On[0]); i++)
    sigaction(Handed_On[i], &sa, NULL);
}

static const char *Usage[] = { " [-v] [-T<int(4)>] [-P<dir(/tmp)>]",
                               " [-o<output>] [-e<int(4)>] <source>[.ktab]" };

int main(int argc, char *argv[])
{ smg_cli c;
  char *OUT, *SRC;
  int   i;
  const double t_start = now_s(), rt_start = real_s();

  Prog_Name = "hetmers";
  smg_cli_parse(argc, argv, &c);
  if (c.argc != 2)
    { fprintf(stderr, "\nUsage: %s %s\n", Prog_Name, Usage[0]);
      fprintf(stderr, "       %*s %s\n", (int) strlen(Prog_Name), "", Usage[1]);
      smg_cli_usage_tail();
    }
  SRC = argv[1];
  OUT = c.out != NULL ? c.out : path_n_root(argv[1], ".ktab");

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

  { smg_opts  opts;
    char *input;
    Shared *sh = NULL;
    /* (a profiler or any other atexit hook -- SMUDGEPLOT_FULL_EXIT -- wants the GPU work in the process it was started with) */
    int   two = getenv("SMUDGEPLOT_ONE_PROCESS") == NULL && getenv("SMUDGEPLOT_FULL_EXIT") == NULL;
    int   to_worker[2] = { -1, -1 }, to_starter[2] = { -1, -1 };
    pid_t pid = -1;

    Load_Lazy = 1;                 /* stub + index only: the engine streams the parts into HBM (smg_ingest.hpp) */
    if (two)
      { sh = (Shared *) mmap(NULL, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        if (sh == MAP_FAILED) { sh = NULL; two = 0; }
      }
    if (two && (pipe(to_worker) != 0 || pipe(to_starter) != 0)) two = 0;
    if (two)
      { const pid_t starter = getpid();
        sh->smu_state = 0;
        fflush(NULL);
        pid = fork();
        if (pid < 0) two = 0;
        if (pid == 0)
          { install(worker_signalled);
            prctl(PR_SET_PDEATHSIG, SIGTERM);                    /* the starter gone (however: SIGKILL too) = this one gone */
            if (getppid() != starter) _exit(1);                  /* (it went between fork() and prctl()) */
          }
      }
    if (two && pid == 0)
      { /* ---- the worker: start the runtime, wait for the table, do the work ---- */
        char go = 0, status;
        close(to_worker[1]); close(to_starter[0]);
        Smu_State = &sh->smu_state;
        (void) smg_device_count();                               /* the first HIP call of the process */
        if (read(to_worker[0], &go, 1) != 1 || go != 1) _exit(1);   /* (the starter failed, and has said why) */
        { smg_ktab T;
          smg_ktab_set_index_memory(sh->index, SHARED_IXWORDS, 1);           /* (the index is there: the stub is only looked at) */
          load_or_die(sh->name, &T);
          smg_ktab_set_index_memory(NULL, 0, 0);
          /* (run() frees the name of a temporary table it was given: a heap copy, not the shared mapping's array) */
          status = (char) run(&c, OUT, &T, &sh->opts, sh->have_input ? strdup(sh->input) : NULL, t_start, rt_start, sh->t_probe);
        }
        fflush(NULL);
        prctl(PR_SET_PDEATHSIG, 0);                              /* the .smu is closed: the starter may leave first from here on */
        if (write(to_starter[1], &status, 1) != 1) _exit(1);
        _exit(status);                                           /* (nobody waits for what this takes) */
      }
    if (two)
      { /* ---- the starter: open and probe the table while the worker's runtime comes up ---- */
        smg_ktab T;
        char go = 1, status = 1;
        close(to_worker[0]); close(to_starter[1]);
        Worker_Pid = pid;
        install(starter_signalled);
        smg_ktab_set_index_memory(sh->index, SHARED_IXWORDS, 0);
        input = smg_cli_open_table(&c, SRC, &T, &opts);          /* (exits 1 with the reference's message when it cannot) */
        smg_ktab_set_index_memory(NULL, 0, 0);
        if (strlen(input ? input : SRC) >= sizeof(sh->name)) { fprintf(stderr, "%s: table name too long\n", Prog_Name); exit(1); }
        sh->opts = opts;
        sh->have_input = input != NULL;
        snprintf(sh->input, sizeof(sh->input), "%s", input ? input : "");
        snprintf(sh->name, sizeof(sh->name), "%s", input ? input : SRC);
        sh->t_probe = now_s();
        if (write(to_worker[1], &go, 1) != 1) { fprintf(stderr, "%s: lost the worker process\n", Prog_Name); exit(1); }
        if (read(to_starter[0], &status, 1) != 1)                /* the worker died without a word: its exit status says how */
          { int ws = 0;
            waitpid(pid, &ws, 0);
            if (sh->smu_state == 1)                              /* it died over the output: no half-written .smu stays behind */
              { char *name = (char *) malloc(strlen(OUT) + 8);
                if (name != NULL) { sprintf(name, "%s.smu", OUT); unlink(name); free(name); }
              }
            if (WIFSIGNALED(ws))
              fprintf(stderr, "%s: the GPU worker process ended unexpectedly (signal %d)\n", Prog_Name, WTERMSIG(ws));
            else
              fprintf(stderr, "%s: the GPU worker process ended unexpectedly\n", Prog_Name);
            exit(1);
          }
        fflush(NULL);
        _exit(status);
      }
    if (sh != NULL) munmap(sh, sizeof(Shared));
    /* ---- one process ---- */
    { smg_ktab T;
      input = smg_cli_open_table(&c, SRC, &T, &opts);
      i = run(&c, OUT, &T, &opts, input, t_start, rt_start, now_s());
    }
  }

  free(OUT);
  /* The result file is closed and the engine has released its device: what exit() would still do is run the atexit
     teardown of the HIP runtime (~95 ms of a 0.57 s run, profiles/r03_e2e_1e9_entries.json).  Flush and leave. */
  fflush(NULL);
  if (getenv("SMUDGEPLOT_FULL_EXIT") != NULL)      /* a profiler or any other atexit hook that must run: exit() as the reference */
    exit(i);
  _exit(i);                                        /* (atexit handlers are skipped: this program registers none) */
}
