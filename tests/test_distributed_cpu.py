"""N > 1 path on CPU: world_size-2 gloo processes shard a small frame by image strips exactly as
bench.py does on GPUs (each rank renders its rows -- here with the oracle standing in for the device
renderer --, one gather of padded RGBA8 payloads to rank 0, de-interleave), and the assembled image must
equal the unsharded frame byte for byte.  With sub > 1 every rank renders its share as `sub` sub-shares -- virtual ranks rank * sub + k of
a world of world * sub -- into one payload [sub][rows][4]: what bench.py --gpus N --sub-shares 2 does (the gathered buffer is then already
in virtual-rank order)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import adanerf_oracle as O
from adanerf_amd import sharding as S

W, H, STRIP = 40, 27, 4      # ragged: 27 rows / 4-row strips -> 7 strips, last one 3 rows


def _scene():
    sc = O.Scene((0.5, -1.0, 1.25), (0.7, 0.7, 0.2), (0.15, 8.25), 1.125, 8.75, 4, 0.2)
    return sc, O.synthetic_weights(5, oracle_bias=0.1, oracle_scale=0.3)


def _render_rows(sc, wts, rows):
    dirs = O.generate_ray_directions(W, H, sc.fov).reshape(H, W, 3)[rows].reshape(-1, 3)
    pose = np.array(sc.view_cell_center, dtype=np.float32)
    res = O.render_rays(dirs, pose, O.camera_rotation(30.0, 5.0), sc, wts, W, H)
    return O.to_rgba8(res["rgb"])


def _worker(rank, world, port, out_path, sub=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, wts = _scene()
    vworld = world * sub
    rmax = S.rays_local_max(W, H, STRIP, vworld)
    payload = torch.zeros((sub, rmax, 4), dtype=torch.uint8)
    for k in range(sub):
        vrank = rank * sub + k
        rgba = _render_rows(sc, wts, S.rows_of_rank(H, STRIP, vworld, vrank))
        assert rgba.shape[0] == S.rays_local(W, H, STRIP, vworld, vrank)
        payload[k, :rgba.shape[0]] = torch.from_numpy(rgba)
    gathered = [torch.zeros_like(payload) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, gathered, dst=0)
    if rank == 0:
        img = S.assemble(torch.stack(gathered).numpy().reshape(vworld, rmax, 4), W, H, STRIP, vworld)
        np.save(out_path, img)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,sub", [(2, 1), (3, 1), (2, 2)])
def test_strip_sharded_frame_over_gloo(tmp_path, world, sub):
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), out, sub), nprocs=world, join=True)
    sc, wts = _scene()
    full = _render_rows(sc, wts, np.arange(H))
    assert np.array_equal(np.load(out), full)


def test_sharding_partition_properties():
    for (w, h, strip, world) in [(800, 800, 8, 8), (1920, 1080, 8, 8), (100, 37, 4, 3), (64, 5, 8, 4), (7, 9, 1, 2)]:
        seen = np.zeros(w * h, dtype=np.int32)
        sizes = []
        for rank in range(world):
            pix = S.local_to_pixel(w, h, strip, world, rank)
            seen[pix] += 1
            sizes.append(pix.size)
            assert pix.size == S.rays_local(w, h, strip, world, rank)
        assert (seen == 1).all()
        assert max(sizes) == S.rays_local_max(w, h, strip, world)
