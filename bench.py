#!/usr/bin/env python3
"""Headline benchmark: training patches/sec of the U-Net step on 128x128x3 patches.

    python bench.py --gpus N --steps 20 --warmup 5          # N > 1: starts one child process per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W   # or under an external launcher

A step = forward + BCE/dice loss + backward + (RCCL gradient all-reduce when N > 1) + global-norm
clip + Adam on one batch of 64 synthetic patches per GPU that is already resident in HBM.
Prints ONE JSON line on rank 0.  --dtype picks the arithmetic of the contractions: f32 (the default line) = float32
by 3 x bf16 splitting, float32-level accuracy -- the arithmetic of the reference's CPU path, which is the parity
target (on CPU the reference's autocast / GradScaler are off, train_model.py:131,144); f32mfma = native
float32 MFMA; bf16 = a BUILDER-CHOSEN reduced-precision mode (bfloat16 activations in HBM and bf16 MFMA operands,
float32 accumulation / BatchNorm / loss / optimiser).  It is NOT the reference's arithmetic: on a GPU the reference
autocasts to float16 with a GradScaler and clips the still-scaled gradients.  At N = 1 the default U-Net line also
carries the bf16 measurement of the same step as the companion object "bfloat16".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: fp32-input MFMA, dense
PEAK_BF16_MFMA_TFLOPS = 2516.0     # v_mfma_f32_32x32x16_bf16, dense
PEAK_HBM_GBS = 8000.0
SUSTAINED_3XBF16_TFLOPS = 272.8    # measured (tools/mfma_shape.hip): float32 work the MFMA + LDS loop sustains under the power cap
# `roofline.peak` is the peak of the INSTRUCTION the dominant kernel issues, in units of the algorithmic FLOPs it
# is credited with: the default float32 path issues six v_mfma_f32_32x32x16_bf16 per 32x32x16 block product
# (3 x bf16 splitting), so its ceiling is 2516 / 6 = 419.3 TFLOP/s of float32 work.
PEAK_BY_DTYPE = {"f32": PEAK_BF16_MFMA_TFLOPS / 6, "f32planes": PEAK_BF16_MFMA_TFLOPS / 6, "f32mfma": PEAK_F32_MFMA_TFLOPS,
                 "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16regs": PEAK_BF16_MFMA_TFLOPS}
INSTR_BY_DTYPE = {"f32": "6 x v_mfma_f32_32x32x16_bf16 per 32x32x16 block product (float32 by 3 x bf16 splitting)",
                  "f32planes": "6 x v_mfma_f32_32x32x16_bf16 per 32x32x16 block product (pre-split plane tensors)",
                  "f32mfma": "v_mfma_f32_32x32x2_f32", "bf16": "v_mfma_f32_32x32x16_bf16",
                  "bf16regs": "v_mfma_f32_32x32x16_bf16 in the 3x3 stride-1 convs (conv_ws.hip, operands rounded at staging); the "
                              "stride-2 / 1x1 / weight-gradient kernels of this mode still issue v_mfma_f32_32x32x8_bf16"}
MODE_BY_DTYPE = {"f32": "float32", "f32mfma": "float32_mfma", "bf16": "bfloat16", "f32planes": "float32_planes",
                 "bf16regs": "bfloat16_regs"}


def usable_cpus():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(workload, features, size, gpu_batch=64, seconds_cap=25.0):
    """The oracle (torch-CPU fp32 restatement of the reference step) timed on this host.  A BOUNDED sample:
    batch 4 (BASELINE configs[0]; batch 1 for the 1024x1024 workload) -- not the GPU's batch 64 -- first on all
    usable host threads, then on ONE thread (SURVEY 8d asks for both).  A reported baseline, not the target."""
    import torch

    from oracle import cnn_ref, resnet_unet_ref, unet_ref
    threads = usable_cpus()
    if workload == "cnn3":
        mk, step, what = cnn_ref.init_state, cnn_ref.train_step, f"cnn_ref SimpleCNN(3,1,{features})"
    elif workload.startswith("resnet"):
        mk, step, what = resnet_unet_ref.init_state, resnet_unet_ref.train_step, f"resnet_unet_ref ResNetUNet(3,1,{features})"
    else:
        mk, step, what = unet_ref.init_state, unet_ref.train_step, f"unet_ref UNet(3,1,{features})"
    batch = 1 if size >= 512 else 4
    g = torch.Generator().manual_seed(0)
    x = torch.randn(batch, 3, size, size, generator=g)
    y = (torch.rand(batch, 1, size, size, generator=g) > 0.8).float()

    def timed(nthreads, warm, n_timed, cap, min_timed=2):
        torch.set_num_threads(nthreads)
        st = mk(3, 1, features, seed=0)
        adam = unet_ref.new_adam_state(st)
        t_all0, times = time.perf_counter(), []
        for i in range(warm + n_timed):
            t0 = time.perf_counter()
            step(st, adam, x, y, lr=1e-4, weight_decay=1e-5)
            dt = time.perf_counter() - t0
            if i >= warm:
                times.append(dt)
            if time.perf_counter() - t_all0 > cap and len(times) >= min_timed:
                break
        return float(np.median(times)), len(times)

    warm, n_timed = (1, 3) if size >= 512 else (2, 10)
    med, n = timed(threads, warm, n_timed, seconds_cap * 0.6)
    med1, n1 = timed(1, 1, 3, seconds_cap * 0.4) if size < 512 else (None, 0)
    unit = "patches/s" if size < 512 else "samples/s"
    out = {"value": round(batch / med, 3), "unit": unit, "cores": threads, "kind": "port",
           "sample": f"oracle/{what}.train_step, batch {batch} (the GPU line is batch {gpu_batch}) x {size}x{size}x3 fp32, "
                     f"{n} timed steps after {warm} warm-up, median {med * 1e3:.1f} ms/step, "
                     f"torch.set_num_threads({threads})"}
    if med1:
        out["single_thread"] = {"value": round(batch / med1, 3), "unit": unit, "cores": 1,
                                "sample": f"same step, torch.set_num_threads(1), {n1} timed steps, "
                                          f"median {med1 * 1e3:.1f} ms/step"}
    if size < 512 and gpu_batch != batch:
        # the same step at the GPU line's own batch: a short sample, all threads
        batch, x_small, y_small = gpu_batch, x, y
        x = torch.randn(batch, 3, size, size, generator=g)
        y = (torch.rand(batch, 1, size, size, generator=g) > 0.8).float()
        medb, nb = timed(threads, 1, 3, 12.0, min_timed=3)
        out["at_gpu_batch"] = {"value": round(batch / medb, 3), "unit": unit, "cores": threads,
                               "sample": f"same step at batch {batch}, {nb} timed steps after 1 warm-up, median "
                                         f"{medb * 1e3:.1f} ms/step, torch.set_num_threads({threads})",
                               "note": "slower per patch than the batch-4 sample: autograd keeps every activation of the step "
                                       "(about 40 MB per 128x128 patch for UNet(3,1,32): ~0.15 GB at batch 4, ~2.5 GB at batch 64), "
                                       "so the batch-64 step streams its working set from DRAM where the batch-4 step largely "
                                       "stays in the host's last-level cache (an estimate from tensor sizes, not a measurement)"}
    return out


def log(msg):
    print(f"[bench +{time.perf_counter() - T0:.1f}s] {msg}", file=sys.stderr, flush=True)


T0 = time.perf_counter()


# ---------------------------------------------------------------------------------------------- multi-GPU supervision
# A run with N > 1 ranks must not be able to end in silence.  The process the caller starts -- `python bench.py --gpus N`
# alone, or one rank of `python -m torch.distributed.run ... bench.py --gpus N` -- is a SUPERVISOR: it never touches the GPU
# or the HIP library, starts the actual rank(s) as fresh child processes (RFI_BENCH_WORKER=1), reads the heartbeat line each
# child writes when it passes a phase (imports, control plane, RCCL init + first all-reduce, first training step, warm-up,
# timed region, done) and bounds every phase.  A child that exits non-zero or overstays a phase ends the attempt: every
# child is killed and a SECOND attempt starts a fresh set of children with RFI_NO_BUCKETS=1 RFI_NO_STOP_EVENTS=1 (ONE
# all-reduce after the backward pass on the main stream, event-record packets between the streams); its JSON line carries
# "fallback": "unbucketed".  A second failure exits non-zero with the phase, the rank and every local rank's stderr tail.
# Under an external launcher the per-rank supervisors agree on "this attempt failed" / "everybody is done" through a
# TCPStore (the launcher's own agent store, or one hosted by rank 0's supervisor).
PHASES = ("spawned", "imports", "control_plane", "rccl_init", "first_step", "warmup", "timed", "done")
# seconds a child may spend getting from the phase named to the next one (the first `import torch` on a fresh box takes 1-2 min)
PHASE_BOUND = {"spawned": 200.0, "imports": 60.0, "control_plane": 60.0, "rccl_init": 60.0, "first_step": 45.0, "warmup": 60.0,
               "timed": 90.0}


def beat(phase):
    """Worker side: one heartbeat line per phase passed (the supervisor's watchdog reads the last one)."""
    path = os.environ.get("RFI_BENCH_HB")
    if path:
        with open(path, "a") as f:
            f.write(f"{phase} {time.time():.3f}\n")
    if os.environ.get("RFI_BENCH_STALL_PHASE") == phase and os.environ.get("RFI_BENCH_STALL_RANK") == os.environ.get("RANK") \
            and not os.environ.get("RFI_BENCH_FALLBACK"):
        time.sleep(3600)          # fault injection (launcher tests): this rank hangs right after `phase`, first attempt only


class Supervisor:
    def __init__(self, n, argv):
        import tempfile
        self.n, self.argv = n, argv
        self.external = "WORLD_SIZE" in os.environ          # one rank of an external launcher: supervise that rank only
        self.ranks = [int(os.environ["RANK"])] if self.external else list(range(n))
        self.limit = float(os.environ.get("RFI_BENCH_LAUNCH_TIMEOUT", "240"))
        self.total = float(os.environ.get("RFI_BENCH_TOTAL_TIMEOUT", "570"))
        self.t0 = time.time()
        pt = os.environ.get("RFI_BENCH_PHASE_TIMEOUT")
        # the override (launcher tests) leaves "spawned" alone: eight cold `import torch` at once take longer than any
        # bound a test would want for the phases after them
        self.bound = {k: (float(pt) if pt and k != "spawned" else v) for k, v in PHASE_BOUND.items()}
        self.tmp = tempfile.mkdtemp(prefix="rfi_bench_")
        self.store = None
        self.procs, self.errs, self.out0 = {}, {}, None
        self.warned = set()

    # ---- coordination between the supervisors of an external launcher's ranks
    def connect_store(self):
        if not self.external:
            return
        import datetime

        from torch.distributed import TCPStore
        addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500"))
        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"     # torch.distributed.run hosts a store there
        self.store = TCPStore(addr, port, None if agent else self.n, is_master=(not agent and self.ranks[0] == 0),
                              timeout=datetime.timedelta(seconds=60), wait_for_workers=False)

    def worker_port(self, attempt):
        import socket
        if not self.external:
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                return sock.getsockname()[1]
        key = f"rfi_bench/port/{attempt}"
        if self.ranks[0] == 0:
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            self.store.set(key, str(port))
            return port
        return int(self.store.get(key))          # (blocks until rank 0's supervisor has published it; store timeout 60 s)

    def flag(self, key, value=None):
        """Set (value given) or test a flag every supervisor sees.  A store that has gone away (its host, the supervisor of
        rank 0 or the launcher's agent, has exited) leaves this supervisor with its local watchdog only."""
        if self.store is None:
            return False
        try:
            if value is not None:
                self.store.set(key, value)
                return True
            return self.store.check([key])
        except Exception:
            self.store = None
            return False

    def count(self, key, inc):
        if self.store is None:
            return None
        try:
            return self.store.add(key, inc)
        except Exception:
            self.store = None
            return None

    # ---- children
    def spawn(self, attempt, port):
        import subprocess
        import tempfile
        self.procs, self.errs = {}, {}
        self.out0 = tempfile.TemporaryFile()
        for r in self.ranks:
            env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}     # the children rendezvous on
            env.update(RANK=str(r), LOCAL_RANK=str(r if not self.external else os.environ.get("LOCAL_RANK", r)),       # their own store
                       WORLD_SIZE=str(self.n), LOCAL_WORLD_SIZE=str(self.n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       RFI_BENCH_WORKER="1", RFI_BENCH_HB=os.path.join(self.tmp, f"hb_{attempt}_{r}"))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if attempt > 0:
                env.update(RFI_NO_BUCKETS="1", RFI_NO_STOP_EVENTS="1", RFI_BENCH_FALLBACK="unbucketed")
            self.errs[r] = tempfile.TemporaryFile()
            self.procs[r] = subprocess.Popen([sys.executable, os.path.abspath(__file__), *self.argv], env=env,
                                             stdout=self.out0 if r == 0 else subprocess.DEVNULL, stderr=self.errs[r])

    def last_beat(self, attempt, r):
        try:
            with open(os.path.join(self.tmp, f"hb_{attempt}_{r}")) as f:
                lines = f.read().split("\n")
            phase, t = [ln for ln in lines if ln][-1].split()
            return phase, float(t)
        except Exception:
            return "spawned", None

    @staticmethod
    def tail(f, nbytes=3000):
        f.seek(0, os.SEEK_END)
        f.seek(max(0, f.tell() - nbytes))
        return f.read().decode(errors="replace")

    def stop_all(self):
        import subprocess
        for p in self.procs.values():
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 5.0
        for p in self.procs.values():
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    def attempt(self, a):
        """Run one set of children to the end.  Returns None on success, else the reason of the failure."""
        port = self.worker_port(a)
        t_start = time.time()
        self.spawn(a, port)
        fail_key, ok_key = f"rfi_bench/fail/{a}", f"rfi_bench/ok/{a}"
        reported_ok, failed = False, None
        # the attempt's limit runs from the moment every local child has finished its imports (a cold `import torch` on a
        # fresh node has its own bound, PHASE_BOUND["spawned"], and says nothing about the exchange); the whole run stays
        # under RFI_BENCH_TOTAL_TIMEOUT (570 s: the driver's limit is 600)
        t_ready = None
        while failed is None:
            now = time.time()
            if t_ready is None and all(self.last_beat(a, r)[0] not in (None, "spawned") for r in self.ranks):
                t_ready = now
            limit = min(self.limit, self.t0 + self.total - (t_ready if t_ready is not None else now))
            codes = {r: p.poll() for r, p in self.procs.items()}
            for r, c in codes.items():
                if c not in (None, 0):
                    if self.last_beat(a, r)[0] == "done":      # everything was measured and printed: a crash in the teardown of
                        codes[r] = 0                           # the process is reported, not turned into a failed run
                        if r not in self.warned:
                            self.warned.add(r)
                            sys.stderr.write(f"bench: rank {r} exited with code {c} AFTER its last phase ('done'); ignored\n")
                        continue
                    failed = f"rank {r} exited with code {c} after phase '{self.last_beat(a, r)[0]}'"
            if failed is None:
                for r, c in codes.items():
                    if c is not None:
                        continue
                    phase, t = self.last_beat(a, r)
                    since = now - (t if t is not None else t_start)
                    if phase in self.bound and since > self.bound[phase]:
                        failed = (f"rank {r} has been in the phase after '{phase}' (-> '{PHASES[PHASES.index(phase) + 1]}') for "
                                  f"{since:.0f} s (bound {self.bound[phase]:.0f} s)")
            if failed is None and t_ready is not None and now - t_ready > limit:
                failed = f"no result {limit:.0f} s after the imports (RFI_BENCH_LAUNCH_TIMEOUT / RFI_BENCH_TOTAL_TIMEOUT)"
            if failed is not None:
                self.flag(fail_key, f"[supervisor of rank {self.ranks[0]}] {failed}")
                break
            if self.flag(fail_key):
                try:
                    failed = "another rank's supervisor ended the attempt: " + self.store.get(fail_key).decode(errors="replace")
                except Exception:
                    failed = "another rank's supervisor ended the attempt"
                break
            if all(c == 0 for c in codes.values()):
                if not reported_ok:
                    self.count(ok_key, 1)
                    reported_ok = True
                done = self.count(ok_key, 0)
                if done is None or done >= self.n:        # every rank's child has finished (or nobody is left to ask)
                    return None
            time.sleep(0.1)
        self.stop_all()
        return failed

    def leave(self):
        """The supervisor that hosts the store leaves last (the others still poll it until they have seen everybody finish)."""
        if self.store is None:
            return
        hosts = self.external and self.ranks[0] == 0 and os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") != "True"
        n = self.count("rfi_bench/bye", 1)
        t_end = time.time() + 10.0
        while hosts and n is not None and n < self.n and time.time() < t_end:
            time.sleep(0.05)
            n = self.count("rfi_bench/bye", 0)

    def run(self):
        try:
            self.connect_store()
        except Exception as e:          # no coordination possible: a local watchdog is still better than none
            sys.stderr.write(f"bench: no store between the supervisors ({e}); watchdog only, no coordinated fallback\n")
            self.store = None
        reasons = []
        for a in range(2):
            why = self.attempt(a)
            if why is None:
                self.leave()
                sys.stderr.write(self.tail(self.errs[self.ranks[0]], 20000))        # progress log of the first local rank
                if 0 in self.ranks:
                    self.out0.seek(0)
                    sys.stdout.write(self.out0.read().decode())
                    sys.stdout.flush()
                return 0
            reasons.append(why)
            sys.stderr.write(f"bench: attempt {a + 1} ({'bucketed exchange' if a == 0 else 'unbucketed fallback'}) failed: {why}\n")
            for r in self.ranks:
                sys.stderr.write(f"---- stderr tail of rank {r}:\n{self.tail(self.errs[r])}\n")
            if a == 0:
                sys.stderr.write("bench: starting a fresh set of ranks with RFI_NO_BUCKETS=1 RFI_NO_STOP_EVENTS=1\n")
            sys.stderr.flush()
        sys.stderr.write("bench: both attempts failed: " + " | ".join(reasons) + "\n")
        return 1


def roofline_of(launches, fam_serial, profile_steps, dtype, workload, peak_tf):
    """Roofline of the dominant MFMA kernel family from the per-launch HIP-event records of the SERIAL profile
    steps (each kernel alone on the chip).  Per launch the attainable time is max(flops / MFMA peak, bytes / HBM
    peak) with the ALGORITHMIC flops and bytes of that launch; `bound` is the roof that accounts for most of the
    family's attainable time and `achieved` / `peak` / `frac` are quoted against that roof; `model_frac` is the
    roofline-model fraction sum(attainable) / sum(measured) over the family's launches."""
    roof = {"bound": "mfma", "achieved": None, "peak": round(peak_tf, 1), "unit": "TFLOP/s", "frac": None,
            "traffic": None, "instruction": INSTR_BY_DTYPE[dtype]}
    mfma_fams = [k for k in ("conv_igemm_mfma", "wgrad_igemm_mfma") if k in fam_serial and fam_serial[k]["ms"]]
    if not mfma_fams:
        return roof, None
    dom = max(mfma_fams, key=lambda k: fam_serial[k]["ms"])
    rows = [r for r in launches if r["family"] == dom and r["ms"] > 0]
    t_meas = sum(r["ms"] for r in rows) * 1e-3
    fl = sum(r["gflop"] for r in rows) * 1e9
    by = sum(r["mbytes"] for r in rows) * 1e6
    t_f = [r["gflop"] * 1e9 / (peak_tf * 1e12) for r in rows]
    t_b = [r["mbytes"] * 1e6 / (PEAK_HBM_GBS * 1e9) for r in rows]
    att_f = sum(a for a, b in zip(t_f, t_b) if a >= b)
    att_b = sum(b for a, b in zip(t_f, t_b) if b > a)
    n_f = sum(1 for a, b in zip(t_f, t_b) if a >= b)
    hbm_bound = att_b > att_f
    if hbm_bound:
        ach = by / t_meas / 1e9
        roof.update(bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4))
    else:
        ach = fl / t_meas / 1e12
        roof.update(achieved=round(ach, 3), frac=round(ach / peak_tf, 4))
    roof.update(kernel=dom,
                mode="serial profile steps (side-stream overlap off, each kernel alone); the timed region runs with "
                     "the weight-gradient kernels overlapped on a side stream",
                avg_launch_ms=round(t_meas * 1e3 / len(rows), 5), launches_per_step=len(rows) / profile_steps,
                algorithmic_flops_per_launch=fl / len(rows), algorithmic_bytes_per_launch=by / len(rows),
                tflops=round(fl / t_meas / 1e12, 3), frac_of_mfma_peak=round(fl / t_meas / 1e12 / peak_tf, 4),
                gbs=round(by / t_meas / 1e9, 1), frac_of_hbm_peak=round(by / t_meas / 1e9 / PEAK_HBM_GBS, 4),
                model_frac=round((att_f + att_b) / t_meas, 4),
                launches_mfma_bound=n_f, launches_hbm_bound=len(rows) - n_f)
    if dtype in ("f32", "f32planes"):    # secondary: the same float32 work against what the native float32 MFMA could do
        roof["vs_native_f32_mfma_peak_157.3"] = round(fl / t_meas / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
        # ... and against what the matrix cores SUSTAIN for this instruction stream under the chip's power cap: the kernels'
        # consumer loop alone (LDS fragment reads + six MFMAs per block product, random data, nothing else on the chip) --
        # tools/mfma_shape.hip, profiles/r4_mfma_shape.txt: 1,637 TFLOP/s of bf16 MFMA at 1.71 GHz = 272.8 of float32 work.
        # A static figure measured once (round 4), not by this run; `peak` stays the datasheet-clock number
        roof["vs_sustained_mfma_rate_272.8"] = round(fl / t_meas / 1e12 / SUSTAINED_3XBF16_TFLOPS, 4)
        roof["sustained_rate_source"] = "static: profiles/r4_mfma_shape.txt (tools/mfma_shape.hip, shape 32 without partner waves)"
    if not roof["frac"] <= 1.0 or not roof["frac_of_mfma_peak"] <= 1.0:
        raise SystemExit(f"roofline fraction {roof['frac']} > 1: wrong peak for the instruction stream")
    # HBM bytes per launch of the dominant family: NOT measured by this run (bench.py cannot sit under the profiler
    # and time itself at once) -- a constant read from the committed rocprofv3 --pmc passes of this same command
    import glob
    tag = {"unet": "", "cnn3": "_cnn3", "unet1024": "_unet1024", "resnet": "_resnet", "resnet1024": "_resnet1024",
           "maskrcnn": "_maskrcnn"}[workload] + \
          ("" if dtype == "f32" else "_" + dtype)
    traffic_files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic{tag}.json")))
    if traffic_files:
        try:
            tj = json.load(open(traffic_files[-1]))["families"].get(dom)
            if tj:
                roof["traffic"] = tj["hbm_bytes_per_launch"]
                roof["traffic_source"] = ("static, from " + os.path.relpath(traffic_files[-1], ROOT) +
                                          " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; "
                                          "not measured by this run)")
        except Exception:
            pass
    return roof, dom


def per_family(report, profile_steps):
    out_ = {}
    for name, f in report.items():
        per = {"launches_per_step": f["launches"] / profile_steps, "ms_per_step": f["ms"] / profile_steps}
        if f["flops"]:
            per["tflops"] = f["flops"] / (f["ms"] * 1e-3) / 1e12 if f["ms"] else None
        if f["bytes"]:
            per["gbs"] = f["bytes"] / (f["ms"] * 1e-3) / 1e9 if f["ms"] else None
        out_[name] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in per.items()}
    return out_


def read_launch_csv(path):
    import csv
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append({"family": r["family"], "label": r["label"], "ms": float(r["ms"]), "gflop": float(r["gflop"]),
                     "mbytes": float(r["mbytes"])})
    return rows


def measure(model, ctx, D, d_x, d_y, hp, args, dtype, rank, launch_csv=None):
    """W untimed + K timed steps of `model` in arithmetic `dtype` (barrier + synchronize on both sides, max over
    ranks), then the per-family HIP-event profile on separate steps."""
    import tempfile
    B, S = args.batch, args.size
    model.set_compute_dtype(MODE_BY_DTYPE[dtype])
    for i in range(args.warmup):
        model.train_step_async(d_x.ptr, d_y.ptr, B, S, S, hp)
        if i == 0:
            ctx.synchronize()                     # the first step (with its first bucketed gradient exchange) has run
            beat("first_step")
    ctx.synchronize()
    D.barrier()
    beat("warmup")
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        model.train_step_async(d_x.ptr, d_y.ptr, B, S, S, hp)
    enq = time.perf_counter() - t0                # host time to enqueue the K steps (the GPU runs behind it)
    ev_ms = ctx.timer_stop()                      # HIP events on the ctx stream; synchronises
    ctx.synchronize()
    wall = time.perf_counter() - t0
    D.barrier()
    wall = D.max_over_ranks(wall)
    beat("timed")
    log(f"[{dtype}] timed region done: {wall * 1e3 / args.steps:.2f} ms/step (host enqueue {enq * 1e3 / args.steps:.2f} ms/step)")
    loss, _ = model.last_loss()
    if not np.isfinite(loss):
        raise SystemExit(f"non-finite loss {loss}")
    # ---- sustained sample: the contract's K steps are a fraction of a second on a power-bound kernel mix; the same step for
    # about two more seconds, in windows of 20 steps (a synchronisation between windows), shows the clock the chip settles at
    sustained = None
    if args.sustain_steps > 0:
        win, times = 20, []
        for _ in range(max(1, args.sustain_steps // win)):
            t1 = time.perf_counter()
            for _ in range(win):
                model.train_step_async(d_x.ptr, d_y.ptr, B, S, S, hp)
            ctx.synchronize()
            times.append(D.max_over_ranks(time.perf_counter() - t1))
        n_s, tot = win * len(times), float(sum(times))
        per = sorted(t * 1e3 / win for t in times)
        sustained = {"steps": n_s, "ms_per_step": round(tot * 1e3 / n_s, 4), "value": round(D.world_size() * B * n_s / tot, 2),
                     "windows_of_20_steps_ms_per_step": {"min": round(per[0], 4), "median": round(per[len(per) // 2], 4), "max": round(per[-1], 4),
                                                         "first": round(times[0] * 1e3 / win, 4), "last": round(times[-1] * 1e3 / win, 4)},
                     "note": "run right after the timed region; `value` of the line stays the contract's K-step figure"}
        log(f"[{dtype}] sustained: {sustained['ms_per_step']:.3f} ms/step over {n_s} steps")
    # ---- per-kernel-family HIP-event profile (separate steps so events do not sit in the timed region).
    # Two passes: OVERLAPPED (the mode the timed region ran in; durations of co-running kernels stretch) and SERIAL
    # (side-stream overlap off: every kernel alone on the chip -> the per-kernel durations the roofline is computed
    # from, comparable with `RFI_NO_OVERLAP=1 rocprofv3 --stats`)
    fam, fam_ov, launches = {}, {}, []
    if args.profile_steps > 0:
        for serial in (False, True):
            ctx.set_overlap(not serial and not os.environ.get("RFI_NO_OVERLAP") == "1")
            ctx.profile_reset()
            ctx.profile(True)
            for _ in range(args.profile_steps):
                model.train_step_async(d_x.ptr, d_y.ptr, B, S, S, hp)
            ctx.synchronize()
            ctx.profile(False)
            if serial:
                fam = ctx.profile_report()
                if rank == 0:
                    path = launch_csv or os.path.join(tempfile.gettempdir(), f"rfi_bench_launches_{os.getpid()}.csv")
                    ctx.profile_dump(path)
                    launches = read_launch_csv(path)
                    if not launch_csv:
                        os.unlink(path)
            else:
                fam_ov = ctx.profile_report()
        ctx.set_overlap(not os.environ.get("RFI_NO_OVERLAP") == "1")
    D.barrier()
    return {"wall": wall, "ev_ms": ev_ms, "loss": float(loss), "fam": fam, "fam_ov": fam_ov, "launches": launches, "sustained": sustained}


def synthetic_instances(batch, size, seed, per_image=3):
    """Patches with `per_image` bright rectangular "emitters" each, and their instance annotations (boxes, labels, masks)."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((batch, size, size, 3)) * 0.1).astype(np.float32)
    targets = []
    for i in range(batch):
        boxes, masks = [], []
        for _ in range(per_image):
            w, h = rng.integers(size // 8, size // 2, 2)
            x1, y1 = rng.integers(0, size - w), rng.integers(0, size - h)
            m = np.zeros((size, size), np.uint8)
            m[y1:y1 + h, x1:x1 + w] = 1
            x[i, y1:y1 + h, x1:x1 + w] += 2.0
            boxes.append([x1, y1, x1 + w, y1 + h]); masks.append(m)
        targets.append({"boxes": np.asarray(boxes, np.float32), "labels": np.ones(per_image, np.int64), "masks": np.stack(masks)})
    return x, targets


def cpu_baseline_maskrcnn(size, gpu_batch, seconds_cap=25.0):
    """The assembled oracle (oracle/mask_rcnn_ref.py: torch-CPU float32, full width) timed on this host: losses + backward
    of one step (no optimiser update) at batch 2 -- a BOUNDED sample of the GPU line's batch."""
    import torch

    from oracle.mask_rcnn_ref import MaskRCNNRef
    threads = usable_cpus()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    ref = MaskRCNNRef(2, 3, 64, 256, 1024)
    batch = 2
    x, targets = synthetic_instances(batch, size, 0)
    t_all0, times = time.perf_counter(), []
    for i in range(1 + 6):
        t0 = time.perf_counter()
        ref.step(x, targets, sampler=(0, i))
        if i >= 1:
            times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all0 > seconds_cap and len(times) >= 2:
            break
    med = float(np.median(times))
    return {"value": round(batch / med, 3), "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"oracle/mask_rcnn_ref MaskRCNNRef(2,3,64,256,1024).step (losses + backward of every parameter set, no optimiser "
                      f"update), batch {batch} (the GPU line is batch {gpu_batch}) x {size}x{size}x3 fp32 with 3 instances per patch, "
                      f"{len(times)} timed steps after 1 warm-up, median {med * 1e3:.0f} ms/step, torch.set_num_threads({threads})"}


def run_maskrcnn(args, ctx, D, rank, local_rank, world, n_ranks_seen):
    """BASELINE configs[3]: one training step of the assembled detector per `step` (backbone, RPN on five levels with its
    losses, proposals, RoI sampling, RoIAlign, box head + Fast R-CNN losses, mask head + mask loss, every backward pass, clip
    + Adam of the four parameter sets).  Tensors AND the box bookkeeping between the stages (top-k, samplers, proposal
    selection, RoI lists) live in HBM; the host reads two RoI counts per step."""
    import torch

    from rfi_toolbox_amd.models import MaskRCNN
    torch.manual_seed(1234)
    B, S = args.batch, args.size
    det = MaskRCNN(2, 3, 64, 256, 1024, device=local_rank, seed=1234 + rank).set_compute_dtype(MODE_BY_DTYPE[args.dtype])
    det.grad_sync = world
    x, targets = synthetic_instances(B, S, 1234 + rank)
    x = ctx.to_device(x)                              # inputs resident in HBM before the timed region (the instance masks go up
    log(f"detector built, batch {B} x {S}x{S}x3 with {len(targets[0]['boxes'])} instances per patch")     # with the first warm-up step)
    losses = None
    for i in range(args.warmup):
        losses = det.train_step(x, targets, masks_resident=True)
        if i == 0:
            ctx.synchronize()
            beat("first_step")
    ctx.synchronize()
    D.barrier()
    beat("warmup")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = det.train_step(x, targets, masks_resident=True)
    ctx.synchronize()
    wall = time.perf_counter() - t0
    D.barrier()
    wall = D.max_over_ranks(wall)
    beat("timed")
    log(f"[{args.dtype}] timed region done: {wall * 1e3 / args.steps:.1f} ms/step")
    if not np.isfinite(losses["loss"]):
        raise SystemExit(f"non-finite loss {losses}")
    fam, launches, host_ms = {}, [], None
    if args.profile_steps > 0:
        import tempfile
        ctx.set_overlap(False)
        ctx.profile_reset()
        ctx.profile(True)
        for _ in range(args.profile_steps):
            det.train_step(x, targets, masks_resident=True)
        ctx.synchronize()
        ctx.profile(False)
        fam = ctx.profile_report()
        if rank == 0:
            path = args.launch_csv or os.path.join(tempfile.gettempdir(), f"rfi_bench_launches_{os.getpid()}.csv")
            ctx.profile_dump(path)
            launches = read_launch_csv(path)
            if not args.launch_csv:
                os.unlink(path)
        ctx.set_overlap(not os.environ.get("RFI_NO_OVERLAP") == "1")
    if world > 1:
        ctx.synchronize()
        D.barrier()
        ctx.comm_destroy()
    if rank != 0:
        beat("done")
        D.shutdown()
        return
    P = max(args.profile_steps, 1)
    ms_per_step = wall * 1e3 / args.steps
    peak = PEAK_BY_DTYPE[args.dtype]
    roof, _ = roofline_of(launches, fam, P, args.dtype, "maskrcnn", peak)
    kernel_ms = sum(f["ms"] for f in fam.values()) / P
    step_flops = sum(f["flops"] for f in fam.values()) / P
    out = {"metric": "training patches/sec (128x128x3)", "value": round(world * B * args.steps / wall, 2), "unit": "patches/s",
           "n_gpus": world, "n_ranks_seen": n_ranks_seen, "fallback": os.environ.get("RFI_BENCH_FALLBACK"), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype.startswith("bf16") else "f32",
           "data": "synthetic",
           "config": {"workload": f"MaskRCNN(2 classes; ResNet-50-FPN 64/256, RPN on P2..P6, RoIAlign 7x7 + 14x14, 1024-wide box head, "
                                  f"4-conv mask head) train step (five losses, every backward pass, clip + Adam of four parameter "
                                  f"sets), batch {B}/GPU x {S}x{S}x3 NHWC with 3 instances per patch (BASELINE configs[3]; "
                                  "builder-defined detector, SURVEY 8a A11)",
                      "arithmetic": ARITHMETIC[args.dtype], "global_batch": B * world, "patch": [S, S, 3], "parallelism": f"dp{world}",
                      "params": int(sum(m.num_parameters() for m in det.models()))},
           "roofline": roof,
           "step": {"algorithmic_gflop_per_patch": round(step_flops / B / 1e9, 3),
                    "tflops_whole_step": round(step_flops / (ms_per_step * 1e-3) / 1e12, 3),
                    "kernel_ms_per_step_serial": round(kernel_ms, 3),
                    "host_bookkeeping_ms_per_step": round(max(ms_per_step - kernel_ms, 0.0), 3),
                    "note": "the box bookkeeping between the stages (top-k, samplers, proposal selection, RoI lists) runs on the "
                            "device; host_bookkeeping = max(0, ms_per_step - kernel_ms_per_step_serial), where the serial sum "
                            "counts the side-stream weight gradients one after the other",
                    "final_losses": {k: round(float(v), 5) for k, v in losses.items()}},
           "families": per_family(fam, P)}
    if world == 1 and not args.no_cpu_baseline:
        log("cpu baseline ...")
        out["cpu_baseline"] = cpu_baseline_maskrcnn(S, B)
    print(json.dumps(out), flush=True)
    beat("done")
    D.shutdown()


ARITHMETIC = {"f32": "float32 (contractions by 3 x bf16 splitting, float32-level accuracy; --dtype f32mfma selects the "
                     "native float32 MFMA)",
              "f32planes": "float32 (3 x bf16 pieces, pre-split plane tensors, LDS-DMA staging)",
              "f32mfma": "float32 (native v_mfma_f32_32x32x2_f32)",
              "bf16": "bfloat16 activations in HBM and bf16 MFMA operands, float32 accumulate, float32 BatchNorm / loss / "
                      "optimiser state -- a builder-chosen reduced-precision mode, NOT the reference's arithmetic (its CPU "
                      "path is float32; on a GPU it autocasts to float16 with a GradScaler, train_model.py:131,144)",
              "bf16regs": "bfloat16 MFMA operands rounded in registers, float32 storage"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("unet", "cnn3", "unet1024", "resnet", "resnet1024", "maskrcnn"), default="unet",
                    help="unet: UNet(3,1,32) batch 64 x 128^2 (headline, BASELINE configs[1]/[4] shape); "
                         "cnn3: the builder-defined 3-layer CNN of configs[1] (SURVEY 8a A9), batch 64 x 128^2; "
                         "unet1024: UNet(3,1,32) on 1 x 1024^2 (configs[2] shape on the reference's U-Net); "
                         "resnet / resnet1024: the builder-defined U-Net with a ResNet-18-style encoder (configs[2], "
                         "SURVEY 8a A10) at 64 x 128^2 / 1 x 1024^2; "
                         "maskrcnn: the builder-defined Mask R-CNN (ResNet-50-FPN, RPN, RoIAlign, box and mask heads; configs[3], "
                         "SURVEY 8a A11) on 64 x 128^2 patches with synthetic instance annotations")
    ap.add_argument("--batch", type=int, default=None, help="patches per GPU per step")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--features", type=int, default=None)
    ap.add_argument("--dtype", choices=("f32", "f32mfma", "bf16", "f32planes", "bf16regs"), default=None,
                    help="f32 (default for unet / cnn3 / resnet): float32 contractions by 3 x bf16 splitting (float32-level "
                         "accuracy, six bf16 MFMAs per product block; the arithmetic of the reference's CPU path); bf16 "
                         "(default for unet1024 / resnet1024, whose BASELINE config names bf16): bfloat16 activations in HBM "
                         "and bf16 MFMA operands, float32 accumulate / BatchNorm / loss / optimiser -- builder-chosen reduced "
                         "precision, not the reference's arithmetic; f32mfma: native float32 MFMA; "
                         "f32planes: the f32 arithmetic on pre-split plane tensors; bf16regs: bf16 operands rounded in "
                         "registers, float32 storage (round 1's bf16 mode)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--sustain-steps", type=int, default=None,
                    help="steps of the sustained sample run after the timed region (default: 300 for the 128x128 workloads, 0 otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16-line", "--no-float32-line", dest="no_companion", action="store_true",
                    help="skip the bfloat16 measurement the default N = 1 U-Net line carries next to the float32 one")
    ap.add_argument("--launch-csv", default=None, help="write the per-launch HIP-event profile here")
    ap.add_argument("--dry-run", action="store_true",
                    help="control plane only (rendezvous, barrier, max-over-ranks, one JSON line); no GPU work")
    args = ap.parse_args()

    if args.gpus > 1 and not os.environ.get("RFI_BENCH_WORKER"):
        raise SystemExit(Supervisor(args.gpus, sys.argv[1:]).run())
    if os.environ.get("RFI_BENCH_FAIL_RANK") is not None and os.environ.get("RFI_BENCH_FAIL_RANK") == os.environ.get("RANK"):
        raise SystemExit("fault injection (RFI_BENCH_FAIL_RANK): this rank exits at init")     # launcher tests
    if args.gpus > 1:
        import torch  # noqa: F401  (the slow import of a fresh box belongs to the phase in front of "imports")
    beat("imports")

    if args.batch is None:
        args.batch = 1 if args.workload.endswith("1024") else 64
    if args.size is None:
        args.size = 1024 if args.workload.endswith("1024") else 128
    if args.features is None:
        args.features = 64 if args.workload in ("cnn3", "resnet", "resnet1024") else 32
    if args.sustain_steps is None:
        args.sustain_steps = 300 if args.workload in ("unet", "cnn3") else 0
    default_dtype = args.dtype is None
    if default_dtype:
        args.dtype = "bf16" if args.workload in ("unet1024", "resnet1024") else "f32"    # (configs[2] names bf16)
    if args.workload == "maskrcnn" and args.dtype in ("bf16", "f32planes"):
        args.dtype = {"bf16": "bf16regs", "f32planes": "f32"}[args.dtype]      # (the detector's models hold float32 tensors)
    if args.workload.startswith("resnet") and (args.dtype == "f32planes" or (args.dtype == "bf16" and args.features % 16)):
        # this model has the bfloat16 data flow only, for widths in whole 16-channel chunks: else operands rounded in registers
        args.dtype = {"bf16": "bf16regs", "f32planes": "f32"}[args.dtype]

    from rfi_toolbox_amd import distributed as D
    rank, local_rank, world = D.init_control_plane("gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    D.barrier()
    beat("control_plane")
    if args.dry_run:
        t0 = time.perf_counter()
        D.barrier()
        wall = D.max_over_ranks(time.perf_counter() - t0)
        seen = D.count_ranks()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "n_ranks_seen": seen, "steps": args.steps, "warmup": args.warmup,
                              "barrier_ms": round(wall * 1e3, 3), "fallback": os.environ.get("RFI_BENCH_FALLBACK")}), flush=True)
        beat("done")
        D.shutdown()
        return

    import torch

    from rfi_toolbox_amd._lib import Hyper
    from rfi_toolbox_amd.data_generation import make_training_patches_device
    from rfi_toolbox_amd.models import SimpleCNN, UNet, UNetResNet18
    from rfi_toolbox_amd.runtime import Context

    log("imports done")
    ctx = Context.get(local_rank)
    D.init_gradient_exchange(ctx, rank, world)
    n_ranks_seen = D.count_ranks_rccl(ctx, world)      # a sum of ones over the RCCL communicator itself
    if n_ranks_seen != world:
        raise SystemExit(f"RCCL communicator sees {n_ranks_seen} ranks, expected {world}")
    beat("rccl_init")

    if args.workload == "maskrcnn":
        return run_maskrcnn(args, ctx, D, rank, local_rank, world, n_ranks_seen)

    def build_model():
        torch.manual_seed(1234)                   # identical replicas on every rank
        if args.workload == "cnn3":
            m = SimpleCNN(3, 1, args.features, device=local_rank)
        elif args.workload.startswith("resnet"):
            m = UNetResNet18(3, 1, args.features, device=local_rank)
        else:
            m = UNet(3, 1, args.features, device=local_rank)
        return m.train()

    model = build_model()
    log("model built")
    B, S = args.batch, args.size
    # synthetic waterfalls -> views/tiling -> 3-channel patches + labels, generated and kept in HBM
    d_x, d_y = make_training_patches_device(B, S, seed=1234 + rank, device=local_rank)
    hp = Hyper(1e-4, 0.9, 0.999, 1e-8, 1e-5, 1.0)   # train_model.py:89,95,130,149 defaults
    log(f"inputs resident: {d_x.shape} {d_y.shape}, label fraction {d_y.numpy().mean():.3f}")

    res = measure(model, ctx, D, d_x, d_y, hp, args, args.dtype, rank, args.launch_csv)
    second = None
    if world == 1 and default_dtype and args.workload == "unet" and not args.no_companion:
        del model
        model2 = build_model()                    # the same step from the same initial weights in the bf16 mode
        second = measure(model2, ctx, D, d_x, d_y, hp, args, "bf16", rank)
        fwd_flops, step_flops = model2.algorithmic_flops(B, S, S)
        n_params = model2.num_parameters()
    else:
        fwd_flops, step_flops = model.algorithmic_flops(B, S, S)
        n_params = model.num_parameters()

    if world > 1:                     # leave the RCCL communicator in an orderly way (every rank, before anyone exits)
        ctx.synchronize()
        D.barrier()
        ctx.comm_destroy()
    if rank != 0:
        beat("done")
        D.shutdown()
        return

    def line_of(r, dtype):
        ms_per_step = r["wall"] * 1e3 / args.steps
        peak = PEAK_BY_DTYPE[dtype]
        roof, _ = roofline_of(r["launches"], r["fam"], max(args.profile_steps, 1), dtype, args.workload, peak)
        step_tflops = step_flops / (ms_per_step * 1e-3) / 1e12
        return {"value": round(world * B * args.steps / r["wall"], 2), "ms_per_step": round(ms_per_step, 4),
                "dtype": "bf16" if dtype.startswith("bf16") else "f32", "arithmetic": ARITHMETIC[dtype],
                "roofline": roof,
                "sustained": r["sustained"],
                "step": {"algorithmic_gflop_per_patch": round(step_flops / B / 1e9, 3),
                         "tflops_whole_step": round(step_tflops, 3),
                         "frac_of_instruction_peak": round(step_tflops / peak, 4),
                         "hip_event_ms_per_step": round(r["ev_ms"] / args.steps, 4), "final_loss": round(r["loss"], 6)},
                "families": per_family(r["fam"], max(args.profile_steps, 1)),
                "families_overlapped": per_family(r["fam_ov"], max(args.profile_steps, 1))}

    main_line = line_of(res, args.dtype)
    out = {
        "metric": ("training patches/sec (128x128x3)" if S == 128 else f"training samples/sec ({S}x{S}x3)"),
        "value": main_line["value"], "unit": "patches/s" if S < 512 else "samples/s",
        "n_gpus": world, "n_ranks_seen": n_ranks_seen, "fallback": os.environ.get("RFI_BENCH_FALLBACK"), "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_line["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": main_line["dtype"], "data": "synthetic",
        "config": {"workload": {
            "unet": f"UNet(3,1,{args.features}) train step (fwd+BCE/dice+bwd+clip+Adam), "
                    f"batch {B}/GPU x {S}x{S}x3 NHWC, BASELINE configs[1] shape on the "
                    "reference's U-Net (the '3-layer CNN' of configs[1] is not in the reference; "
                    "its builder-defined form runs with --workload cnn3)",
            "cnn3": f"SimpleCNN(3,1,{args.features}) = Conv3x3+ReLU, Conv3x3+ReLU, Conv1x1 (BASELINE configs[1]; "
                    f"builder-defined, SURVEY 8a A9) train step, batch {B}/GPU x {S}x{S}x3 NHWC fp32",
            "unet1024": f"UNet(3,1,{args.features}) train step on {B} x {S}x{S}x3 per GPU "
                        "(BASELINE configs[2] shape on the reference's U-Net)",
            "resnet": f"UNetResNet18(3,1,{args.features}) (U-Net with a ResNet-18-style encoder; builder-defined, SURVEY 8a A10) "
                      f"train step, batch {B}/GPU x {S}x{S}x3 NHWC",
            "resnet1024": f"UNetResNet18(3,1,{args.features}) train step on {B} x {S}x{S}x3 per GPU (BASELINE configs[2]; "
                          "builder-defined model, SURVEY 8a A10)"}[args.workload],
                   "arithmetic": main_line["arithmetic"],
                   "global_batch": B * world, "patch": [S, S, 3], "parallelism": f"dp{world}",
                   "params": n_params},
        "roofline": main_line["roofline"],
        "sustained": main_line["sustained"],
        "step": main_line["step"],
        "families": main_line["families"],
        "families_overlapped": main_line["families_overlapped"],
    }
    if second is not None:
        out["bfloat16"] = line_of(second, "bf16")
        out["bfloat16"]["note"] = ("the same step, same inputs and initial weights, with --dtype bf16: a builder-chosen "
                                   "reduced-precision mode (bf16 storage and MFMA operands), NOT the reference's arithmetic and "
                                   "not the parity target; the top-level line is the float32 one")
    if world == 1 and not args.no_cpu_baseline:
        log("cpu baseline ...")
        out["cpu_baseline"] = cpu_baseline(args.workload, args.features, S, B)
    print(json.dumps(out), flush=True)
    beat("done")
    D.shutdown()


if __name__ == "__main__":
    main()
