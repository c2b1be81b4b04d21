"""One rank of the two-process data-parallel parity test (started by tests/test_gpu_z_dp.py, not
collected by pytest).  Runs the HIP UNet on this rank's half of the `dp` golden case over a gloo
process group (both ranks on cuda:0; gloo moves HIP tensors through host memory) and saves what
the parent compares:  python tests/dp_worker.py RANK WORLD PORT OUT_DIR"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import datagen as dg            # noqa: E402
from oracle import unet_oracle as orc       # noqa: E402

CASE = dict(B=4, cin=1, ncls=4, hw=32, seed=81)      # tests/golden/make_golden.py::gold_dp
NAMES = ["a", "b", "c"]


def main():
    if os.environ.get("MIS_DP_DEBUG"):      # a hang prints every thread's stack and ends the rank
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MIS_DP_DEBUG"]), exit=True)
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from medical_image_segmentation_amd.model.unet import UNet
    from medical_image_segmentation_amd.parallel import GradSync
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.metrics import Metrics
    dev = torch.device("cuda:0")
    c = CASE
    per = c["B"] // world
    sl = slice(rank * per, (rank + 1) * per)
    x = torch.from_numpy(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100)[sl]).float().to(dev)
    t = torch.from_numpy(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300)[sl]).to(dev)
    st = orc.make_state(c["cin"], c["ncls"], base=8, seed=c["seed"])
    out = {}
    for mode in ("per_rank", "global"):
        m = UNet(c["cin"], c["ncls"], 0, base_channels=8, compute_dtype=torch.float32)
        m.load_state_dict({k: v.clone() for k, v in st.items()})
        if rank != 0:                      # the broadcast must make the replicas identical
            with torch.no_grad():
                for p in m.parameters():
                    p.mul_(1.5)
        m = m.to(dev).train()
        sync = GradSync(m)
        sync.broadcast_parameters()
        glob = mode == "global"
        loss_fn, metric = diceloss(NAMES, global_batch=glob), Metrics(NAMES, global_batch=glob)
        for _ in range(2):                 # twice: the second step reuses the bucket views of the first
            m.zero_grad(set_to_none=False)
            logits = m(x)
            ld = loss_fn(logits, t)
            mat = metric.update(logits, t)
            sync.begin_step()
            ld["total_loss"].backward()
            sync.finish_step()
        torch.cuda.synchronize()
        out[mode] = {"losses": [float(v.detach()) for v in ld.values()], "metrics": torch.from_numpy(mat),
                     "logits": logits.detach().float().cpu(),
                     "grads": [p.grad.detach().cpu().clone() for p in m.parameters()]}
        sync.remove()
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
