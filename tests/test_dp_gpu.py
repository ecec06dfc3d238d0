"""The RCCL path of image_segmentation_amd.dp.LoRAGradReducer on the GPU (backend "nccl" IS RCCL on ROCm;
train_lora_sam.py:32,126,172).  A one-rank process group is all a 1-GPU box offers, and the reducer issues its collectives
whenever a process group exists, whatever its size: communicator set-up, `broadcast_params`, one asynchronous all-reduce per
LoRA block started from inside the trunk's backward, and `reduce()`.  The mean over one rank is the identity, so the
gradients must equal those of the plain step bit for bit.

 * test_reducer_on_rccl_single_rank: in this process; the reducer's own counters say how many collectives were issued.
 * test_bench_under_launcher_runs_rccl: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` as a child
   process (the way the driver launches N > 1), with RCCL's own log (NCCL_DEBUG=INFO, subsystems INIT + COLL) as the
   witness: communicator initialised and >= 4 AllReduce calls logged by the library.
 * test_plain_bench_line: plain `python bench.py --gpus 1` still prints its one JSON line (no process group, no collectives).
More ranks need more GPUs than the test box has: tests/test_dp_cpu.py covers the arithmetic on two gloo ranks."""
import json
import os
import re
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

from conftest import REPO, TINY

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, env, timeout):
    r = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, f"{' '.join(cmd)} -> rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stdout + "\n" + r.stderr


@pytest.mark.timeout(600)
def test_bench_under_launcher_runs_rccl(tmp_path):
    log = tmp_path / "rccl.%h.%p.log"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,COLL",
               NCCL_DEBUG_FILE=str(log))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--no-cpu-baseline"]
    line, out = _run(cmd, env, 540)
    assert line["n_gpus"] == 1 and line["process_group"].startswith("nccl")
    c = line["collectives_issued"]           # (1 warm-up + 2 timed) steps x 2 LoRA blocks, all from inside the backward
    assert c["broadcast"] == 1 and c["all_reduce"] == 6 and c["all_reduce_from_backward"] == 6, c
    text = out + "".join(p.read_text(errors="replace") for p in tmp_path.iterdir())
    assert re.search(r"NCCL INFO .*(Init COMPLETE|ncclCommInitRank|comm 0x)", text), text[-3000:]
    n_ar = len(re.findall(r"NCCL INFO AllReduce:", text))
    assert n_ar >= 4, f"{n_ar} AllReduce calls in RCCL's log\n{text[-3000:]}"
    print(f"RCCL log: communicator initialised, {n_ar} AllReduce calls")


@pytest.mark.timeout(600)
def test_plain_bench_line():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    line, _ = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
                    "--no-cpu-baseline"], env, 540)
    assert line["n_gpus"] == 1 and line["process_group"] is None
    assert line["collectives_issued"] == {"broadcast": 0, "all_reduce": 0, "all_reduce_from_backward": 0}
    r = line["roofline"]
    assert r["all_timed_kernel_ms_per_step_net"] <= r["profiled_step_wall_ms"]


@pytest.mark.timeout(300)
def test_reducer_on_rccl_single_rank():
    from image_segmentation_amd.dp import LoRAGradReducer
    from image_segmentation_amd.synth import synth_frames
    from test_trunk_gpu import fpn_loss, make_encoder
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    img = synth_frames(2, 256, 0, device="cuda")
    stats = {}

    def step(use_reducer):
        enc = make_encoder(TINY)
        lora = [p for p in enc.parameters() if p.requires_grad]
        red = LoRAGradReducer(lora).attach(enc.trunk) if use_reducer else None
        if red is not None:
            assert red.active()
            red.broadcast_params(0)
        out = []
        for _ in range(2):                       # two steps: the buckets are reused
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = fpn_loss(enc(img))
            loss.backward()
            if red is not None:
                red.reduce()
            torch.cuda.synchronize()
            out.append([p.grad.clone() for p in lora])
            for p in lora:
                p.grad = None
        if red is not None:
            stats.update(red.stats)
        return out

    plain = step(False)                          # before the group exists
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        reduced = step(True)
    finally:
        dist.destroy_process_group()
    # 2 steps x 2 LoRA blocks, every one started by the backward's bucket hook; one parameter broadcast
    assert stats == {"broadcast": 1, "all_reduce": 4, "all_reduce_from_backward": 4}, stats
    for a, b in zip(plain, reduced):
        assert len(a) == 8
        for x, y in zip(a, b):
            assert torch.isfinite(y).all() and torch.equal(x, y)
