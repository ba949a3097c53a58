"""Worker of tests/test_zy_rccl_one_rank.py: runs in its own process so that torch (and with it
torch's HIP runtime) is loaded before libluma_hip.so, the order bench.py uses."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED = 0x4C554D41


def main():
    import torch
    import torch.distributed as dist
    from raytracer_amd import abi, bands

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    bands.FORCE_COLLECTIVE = True

    W, H = 96, 60
    kw = dict(spp=2, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    params = abi.make_params(W, H, **kw)
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.default_scene())
        ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame(W, H)
        want, _ = ctx.render(params)

        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            band = torch.empty((H, W), dtype=torch.int32, device=dev)
            ctx.reset()
            ctx.render_device(params, band.data_ptr(), stream.cuda_stream)
            frame = bands.gather_frame(band, H, W, 0, 1)
            striped = bands.gather_striped(band, H, W, 0, 1, 15)
            stream.synchronize()
            assert np.array_equal(frame.cpu().numpy().view(np.uint32), want), "gather_frame"
            assert np.array_equal(striped.cpu().numpy().view(np.uint32), want), "gather_striped"

            # two frames through the double-buffered gather: frame 0 of a fresh accumulation == want
            pg = bands.PipelinedGather(H, W, 0, 1, 0, dev)
            assert pg.collective
            ctx.reset()
            first = None
            for i in range(2):
                buf = pg.acquire(i)
                ctx.render_device(params, buf.data_ptr(), stream.cuda_stream)
                pg.submit(i)
                if i == 0:
                    pg.works[0].wait()
                    first = pg.gathered[0].clone()
            last = pg.finish()
            stream.synchronize()
            assert np.array_equal(first.reshape(H, W).cpu().numpy().view(np.uint32), want), "pipelined gather"
            assert tuple(last.shape) == (H, W)
            assert ctx.frame_index() == 2 * kw["spp"]
    dist.destroy_process_group()
    print("RCCL_ONE_RANK_OK")


if __name__ == "__main__":
    main()
