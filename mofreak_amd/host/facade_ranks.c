/* facade_ranks N <video_dir> <mofreak_dir>: the dataset loop of computeMoFREAKFiles (src/MoFREAK/main.cpp:854-924) over N
 * GPUs.  Starts N rank processes -- `facade_main files-rank <rank> N <id_file> <video_dir> <mofreak_dir>`, one per GPU -- and
 * waits for them; rank 0 makes the RCCL id and leaves it in <id_file> for the others.  Plain C, libc only: a launcher must
 * not have a GPU runtime loaded when it starts other programs. */
#include <libgen.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

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
    char self[PATH_MAX], prog[PATH_MAX + 32], id_file[PATH_MAX + 64];
    const ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    if (n <= 0) return 2;
    self[n] = 0;
    snprintf(prog, sizeof prog, "%s/facade_main", dirname(self));  /* next to this launcher */
    mkdir(argv[3], 0777);
    snprintf(id_file, sizeof id_file, "%s/.mofreak_rccl_id.%ld", argv[3], (long)getpid());
    pid_t pid[64];
    int started = 0;
    for (int r = 0; r < world; ++r) {
        const pid_t p = fork();
        if (p < 0) break;
        if (p == 0) {
            char rs[16], ws[16];
            snprintf(rs, sizeof rs, "%d", r);
            snprintf(ws, sizeof ws, "%d", world);
            execl(prog, prog, "files-rank", rs, ws, id_file, argv[2], argv[3], (char *)NULL);
            _exit(127);
        }
        pid[started++] = p;
    }
    int rc = started == world ? 0 : 1;
    for (int i = 0; i < started; ++i) {
        int st = 0;
        if (waitpid(pid[i], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : (WIFEXITED(st) && WEXITSTATUS(st) ? WEXITSTATUS(st) : 1);
    }
    unlink(id_file);
    return rc;
}
