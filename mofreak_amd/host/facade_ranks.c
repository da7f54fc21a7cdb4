/* facade_ranks N <video_dir> <mofreak_dir>: the dataset loop of computeMoFREAKFiles (src/MoFREAK/main.cpp:854-924) over N
 * GPUs.  Starts N rank processes -- `facade_main files-rank <rank> N <id_file> <video_dir> <mofreak_dir>`, one per GPU -- and
 * waits for them; rank 0 makes the RCCL id and leaves it in <id_file> for the others.  Plain C, libc only: a launcher must
 * not have a GPU runtime loaded when it starts other programs.
 *
 * A rank that fails must not leave the others waiting in a collective for ever: children are reaped in the order they end;
 * the first one that ends badly (or a fork that fails, or the deadline MOFREAK_RANKS_TIMEOUT_S, default 3600 s) stops the
 * rest -- SIGTERM, two seconds, SIGKILL -- and the launcher returns non-zero. */
#include <errno.h>
#include <libgen.h>
#include <limits.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void stop_all(const pid_t *pid, const int *alive, int n)
{
    for (int i = 0; i < n; ++i)
        if (alive[i]) kill(pid[i], SIGTERM);
    const double until = now_s() + 2.0;
    for (;;) {  /* give them two seconds to go by themselves */
        int left = 0;
        for (int i = 0; i < n; ++i)
            if (alive[i] && kill(pid[i], 0) == 0) ++left;
        if (!left || now_s() > until) break;
        usleep(20000);
        while (waitpid(-1, NULL, WNOHANG) > 0) {
        }
    }
    for (int i = 0; i < n; ++i)
        if (alive[i]) kill(pid[i], SIGKILL);
    while (waitpid(-1, NULL, 0) > 0) {
    }
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: facade_ranks N <video_dir> <mofreak_dir>\n");
        return 2;
    }
    const int world = atoi(argv[1]);
    if (world < 1 || world > 64) {
        fprintf(stderr, "N must be in 1..64\n");
        return 2;
    }
    const char *prog_override = getenv("MOFREAK_RANK_PROGRAM"); /* tests: another program in place of facade_main */
    const char *tmo = getenv("MOFREAK_RANKS_TIMEOUT_S");
    const double deadline = now_s() + (tmo && atof(tmo) > 0 ? atof(tmo) : 3600.0);
    char self[PATH_MAX], prog[PATH_MAX + 32], id_file[PATH_MAX + 64];
    const ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    if (n <= 0) return 2;
    self[n] = 0;
    if (prog_override)
        snprintf(prog, sizeof prog, "%s", prog_override);
    else
        snprintf(prog, sizeof prog, "%s/facade_main", dirname(self)); /* next to this launcher */
    mkdir(argv[3], 0777);
    snprintf(id_file, sizeof id_file, "%s/.mofreak_rccl_id.%ld", argv[3], (long)getpid());
    pid_t pid[64];
    int alive[64];
    int started = 0, rc = 0;
    for (int r = 0; r < world; ++r) {
        const pid_t p = fork();
        if (p < 0) {
            fprintf(stderr, "facade_ranks: fork of rank %d failed: %s\n", r, strerror(errno));
            rc = 1;
            break;
        }
        if (p == 0) {
            char rs[16], ws[16];
            snprintf(rs, sizeof rs, "%d", r);
            snprintf(ws, sizeof ws, "%d", world);
            execl(prog, prog, "files-rank", rs, ws, id_file, argv[2], argv[3], (char *)NULL);
            _exit(127);
        }
        pid[started] = p;
        alive[started++] = 1;
    }
    int left = started;
    while (!rc && left > 0) {
        int st = 0;
        const pid_t p = waitpid(-1, &st, WNOHANG);
        if (p == 0) {
            if (now_s() > deadline) {
                fprintf(stderr, "facade_ranks: deadline passed with %d rank(s) still running\n", left);
                rc = 124;
                break;
            }
            usleep(20000);
            continue;
        }
        if (p < 0) {
            if (errno == EINTR) continue;
            rc = 1;
            break;
        }
        for (int i = 0; i < started; ++i)
            if (pid[i] == p) {
                alive[i] = 0;
                --left;
                if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
                    rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
                    fprintf(stderr, "facade_ranks: rank %d ended with status %d; stopping the others\n", i, rc);
                }
            }
    }
    if (rc) stop_all(pid, alive, started);
    unlink(id_file);
    return rc;
}
