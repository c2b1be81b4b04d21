"""GPU suite: the gradient exchange on a real RCCL communicator (a group of one rank: AVG over one
member is the identity, so a step with the exchange must equal a step without it bit for bit).
Exercises what the CPU gloo tests cannot: collectives issued from autograd's hook thread on the
device, the AVG reduction, waiting on the handles, gradients left as views of the buckets, and the
fused optimizer reading them."""
import os

import pytest
import torch
import torch.distributed as dist

from oracle import datagen as dg
from oracle import unet_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_grad_sync_on_rccl_group_of_one():
    if not torch.cuda.is_available():
        pytest.fail("GPU suite needs a HIP device")
    from medical_image_segmentation_amd.model.unet import UNet
    from medical_image_segmentation_amd.parallel import GradSync
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.optim import clip_and_step
    created = not dist.is_initialized()
    if created:
        # a rendezvous on the loopback interface; a box without a usable port is an environment
        # problem, not a failure of the exchange
        err = None
        for port in (29517, 29618, 29719):
            try:
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                        device_id=torch.device(DEV))
                err = None
                break
            except Exception as e:      # noqa: BLE001
                err = e
        if err is not None:
            pytest.skip(f"no process group on this box: {err}")
    try:
        loss_fn = diceloss(["a", "b", "c"])
        x = torch.from_numpy(dg.hash_normalish((2, 1, 64, 64), 901)).float().to(DEV)
        y = torch.randint(0, 4, (2, 64, 64), generator=torch.Generator().manual_seed(3)).to(DEV)
        res = []
        for use_sync in (False, True):
            m = UNet(1, 4, 0, base_channels=32, compute_dtype=torch.bfloat16)
            m.load_state_dict({k: v.clone() for k, v in orc.make_state(1, 4, base=32, seed=71).items()})
            m = m.to(DEV).train()
            opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
            sync = GradSync(m, always_reduce=True) if use_sync else None
            if sync is not None:
                sync.broadcast_parameters()
                assert len(sync.buckets) == 4 and sync.active and sync.avg_in_collective
            for _ in range(3):
                opt.zero_grad()
                loss = loss_fn(m(x), y)["total_loss"]
                if sync is not None:
                    sync.begin_step()
                loss.backward()
                if sync is not None:
                    sync.finish_step()
                    p0 = next(m.parameters())
                    assert p0.grad.untyped_storage().data_ptr() == sync.flat[sync.bucket_of[p0]].untyped_storage().data_ptr()
                clip_and_step(opt, list(m.parameters()), 1.0)
            torch.cuda.synchronize()
            res.append([p.detach().clone() for p in m.parameters()])
            if sync is not None:
                sync.remove()
        for a, b in zip(*res):
            assert torch.equal(a, b)
    finally:
        if created:
            dist.destroy_process_group()
