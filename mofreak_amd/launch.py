"""One process per GPU, started from a parent that never touches the GPU.

`python bench.py --gpus N` has to run by itself (no torchrun around it).  A process that has initialised HIP must not
be replaced or forked, so the parent here does no GPU call at all: it picks a rendezvous port, starts N fresh children
of the same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, relays rank 0's
standard output (the one JSON line) and waits.  If any rank fails, the others are stopped (they would otherwise sit in a
barrier for ever) and the parent exits non-zero.  The reference's counterpart is the sequential dataset loop of
computeMoFREAKFiles (src/MoFREAK/main.cpp:854-924): here every rank takes its share of it.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time
from typing import Sequence

RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE")


def launched_by_a_launcher() -> bool:
    """True inside a rank process (ours or torch.distributed.run's)."""
    return all(k in os.environ for k in RANK_ENV)


def free_port() -> int:
    """A port rank 0's store can listen on.  Not simply bind(0): the kernel hands out a port again while connections of the
    job that had it a moment ago are still in TIME_WAIT (back-to-back runs), and a listener without SO_REUSEADDR -- the
    store's -- is then refused.  A port picked at random that takes a plain bind + listen has no such leftovers.  Drawn below
    the kernel's ephemeral range (net.ipv4.ip_local_port_range, 32768 up by default): no outgoing connection of this host is
    given such a port between this probe and the store's bind."""
    import random
    lo, hi = 20000, 32768
    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as f:
            eph_lo = int(f.read().split()[0])
        if eph_lo - 1024 >= 2048:
            lo, hi = max(1024, min(20000, eph_lo - 8192)), eph_lo
    except (OSError, ValueError, IndexError):
        pass
    rng = random.Random(os.getpid() ^ time.monotonic_ns())
    for _ in range(64):
        port = rng.randrange(lo, hi)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            try:
                s.bind(("127.0.0.1", port))
                s.listen(1)
            except OSError:
                continue
            return port
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base: dict | None = None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "MOFREAK_LAUNCHER": "mofreak_amd.launch"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on these hosts
    return env


def spawn_ranks(argv: Sequence[str], world: int, timeout_s: float | None = None, poll_s: float = 0.05) -> int:
    """Run `argv` (a full command line, e.g. [sys.executable, "bench.py", ...]) as `world` rank processes.

    The job's stdout is rank 0's JSON line(s): rank 0's stdout is read here and lines that open with `{` are passed
    on; anything else a library prints there (gloo's "[Gloo] Rank 0 is connected ..." banner) and the other ranks'
    stdout go to stderr.  Returns the job's exit code: 0 iff every rank exited 0."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs: list[subprocess.Popen] = []

    def on_signal(signum, _frame):  # the parent is told to stop: take the ranks along (finally: below)
        raise KeyboardInterrupt(f"signal {signum}")

    old = {}
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            old[sig] = signal.signal(sig, on_signal)
        except ValueError:  # not the main thread
            pass
    try:
        for r in range(world):
            procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, port),
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None,
                                          start_new_session=True))  # its own process group: stopped as a group below
        relay = threading.Thread(target=_relay_json_lines, args=(procs[0].stdout,), daemon=True)
        relay.start()
        t0 = time.monotonic()
        codes: list[int | None] = [None] * world
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                print(f"[launch] rank {bad[0]} exited with code {codes[bad[0]]}; stopping the other ranks", file=sys.stderr)
                break
            if timeout_s is not None and time.monotonic() - t0 > timeout_s:
                print(f"[launch] {world} ranks still running after {timeout_s:.0f} s; stopping them", file=sys.stderr)
                codes = [c if c is not None else 124 for c in codes]
                break
            time.sleep(poll_s)
        if all(c == 0 for c in codes):
            relay.join(timeout=30)  # rank 0 has exited: its pipe is at end of file
        return next((c for c in codes if c not in (None, 0)), 0)
    except KeyboardInterrupt as e:
        print(f"[launch] interrupted ({e}); stopping the ranks", file=sys.stderr)
        return 130
    finally:
        _stop(procs)
        for sig, h in old.items():
            signal.signal(sig, h)


def _relay_json_lines(pipe) -> None:
    for raw in iter(pipe.readline, b""):
        line = raw.decode("utf-8", "replace")
        out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
        out.write(line)
        out.flush()
    pipe.close()


def _stop(procs: Sequence[subprocess.Popen]) -> None:
    """End exactly the process groups started above (never by name or pattern)."""
    alive = [p for p in procs if p.poll() is None]
    for p in alive:
        try:
            os.killpg(p.pid, signal.SIGTERM)
        except (ProcessLookupError, PermissionError):
            pass
    t_end = time.monotonic() + 10.0
    for p in alive:
        try:
            p.wait(timeout=max(0.1, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            p.wait()


def self_launch(world: int, script: str, args: Sequence[str], timeout_s: float | None = None) -> int:
    """Re-run `script args` once per rank (from a parent that must not have made a GPU call)."""
    return spawn_ranks([sys.executable, script, *args], world, timeout_s=timeout_s)
