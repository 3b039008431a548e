"""CPU tests of the host logic (NMS, pad/unpad, blobs, meta_data, pose combine) and of the
multi-process path (gloo, world_size 2): frame sharding + the single all-gather of detections."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from posecnn_amd import config, dist as pdist, fcn

F = np.float32


def test_nms_matches_the_reference_on_golden_sets():
    """survivors AND their order against lib/utils/nms.py itself on 33 seeded sets (tests/golden/nms.npz, written by
    tests/golden/make_nms_golden.py importing the reference): n up to 600, duplicates, equal scores, degenerate boxes."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nms.npz"))
    big = 0
    for k in range(int(d["cases"])):
        dets, thresh, want = d["dets_%d" % k], float(d["thresh_%d" % k]), d["keep_%d" % k].tolist()
        assert fcn.nms(dets, thresh) == want, "case %d (n = %d, thresh %g)" % (k, dets.shape[0], thresh)
        big += dets.shape[0] > 16
    assert big >= 20     # both walks (nested lists for a frame's few boxes, vector OR for large sets) are covered


def test_nms_is_class_aware_and_score_ordered():
    # lib/utils/nms.py:3-32
    dets = np.array([
        [0, 1, 10, 10, 50, 50, 0.9],
        [0, 1, 12, 12, 52, 52, 0.8],   # same class, IoU ~0.82 with row 0 -> suppressed
        [0, 2, 12, 12, 52, 52, 0.7],   # other class -> kept
        [0, 1, 100, 100, 140, 140, 0.95],
        [0, 1, 10, 10, 30, 30, 0.5],   # IoU with row 0 = 441/1681 < 0.5 -> kept
    ], F)
    keep = fcn.nms(dets, 0.5)
    assert keep == [3, 0, 2, 4]
    assert fcn.nms(dets[:0], 0.5) == []


def test_pad_unpad_and_image_blob():
    im = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    p = fcn.pad_im(im, 16)
    assert p.shape == (16, 16, 3) and np.all(p[5:] == 0) and np.array_equal(p[:5, :7], im)
    assert fcn.pad_im(np.zeros((480, 640)), 16).shape == (480, 640)
    assert np.array_equal(fcn.unpad_im(p, 16, orig_shape=(5, 7)), im)
    depth = np.array([[0, 1000], [2000, 65535]], np.uint16)
    blob, blob_d, s = fcn._get_image_blob(im[:2, :2], depth)
    assert blob.shape == (1, 2, 2, 3) and s == 1.0
    assert np.allclose(blob[0, 0, 0], im[0, 0].astype(F) - config.PIXEL_MEANS[0, 0])
    # depth tower: clip(d / 2000, 0, 1) * 255 - means (lib/fcn/test.py:70-74)
    assert np.allclose(blob_d[0, 0, 1], 127.5 - config.PIXEL_MEANS[0, 0])
    assert np.allclose(blob_d[0, 1, 1], 255 - config.PIXEL_MEANS[0, 0])
    # the reference's means are float64 (lib/fcn/config.py:242), so `im_orig -= cfg.PIXEL_MEANS` on a float32 image is
    # float32(double(x) - mean): for every byte value, and NOT the float32 - float32(mean) it is easy to write instead
    assert config.PIXEL_MEANS.dtype == np.float64 and blob.dtype == np.float32 and blob_d.dtype == np.float32
    allv = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    b2, _, _ = fcn._get_image_blob(allv, None)
    want = (allv.astype(np.float64) - config.PIXEL_MEANS).astype(np.float32)
    assert np.array_equal(b2[0].view(np.uint32), want.view(np.uint32))
    wrong = allv.astype(np.float32) - config.PIXEL_MEANS.astype(np.float32)
    assert (b2[0] != wrong).mean() > 0.3


def test_meta_data_layout():
    m = config.make_meta_data(config.DEMO_INTRINSICS, voxel_step=(1, 2, 3), voxel_min=(4, 5, 6))
    assert m.shape == (48,) and m.dtype == np.float32
    assert (m[0], m[2], m[4], m[5], m[8]) == (F(1066.778), F(312.9869), F(1067.487), F(241.3109), 1)
    Kinv = m[9:18].reshape(3, 3).astype(np.float64)
    assert np.allclose(Kinv @ config.DEMO_INTRINSICS, np.eye(3), atol=1e-4)
    assert np.all(m[18:42] == 0) and list(m[42:48]) == [1, 2, 3, 4, 5, 6]


def test_combine_poses_copies_class_quaternion():
    rois = np.array([[0, 3, 0, 0, 10, 10, 5.0], [0, 7, 20, 20, 40, 40, 9.0]], F)
    init = np.array([[1, 0, 0, 0, .1, .2, .3], [1, 0, 0, 0, .4, .5, .6]], F)
    tanh = np.zeros((2, 88), F)
    tanh[0, 12:16] = (.5, -.5, .25, .1)
    tanh[1, 28:32] = (.9, .1, .2, .3)
    r, p, keep = fcn.combine_poses(rois, init, tanh)
    assert keep == [1, 0]
    assert np.allclose(p[0], [.9, .1, .2, .3, .4, .5, .6]) and np.allclose(p[1], [.5, -.5, .25, .1, .1, .2, .3])


def test_shard_range_partitions_frames():
    for n, w in ((128, 8), (16, 1), (10, 4), (3, 8)):
        cover = []
        for r in range(w):
            lo, hi = pdist.shard_range(n, r, w)
            cover += list(range(lo, hi))
        assert cover == list(range(n))
    assert pdist.shard_range(128, 3, 8) == (48, 64)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    r, w, _ = pdist.init_from_env(backend="gloo")
    cap, B = 8, 4
    rows = torch.zeros((cap, 14))
    n = 2 + rank  # rank 0 has 2 detections, rank 1 has 3
    for i in range(n):
        rows[i, 0] = i % B          # local image index
        rows[i, 1] = 10 * rank + i  # class tag to recognise the row
        rows[i, 6] = 1.0
    count = torch.tensor([n], dtype=torch.int32)
    g_rows, g_counts = pdist.all_gather_detections(rows, count, frame_offset=rank * B)
    flat = pdist.flatten_gathered(g_rows, g_counts)
    # the pipelined drain used by bench.py must deliver the same rows, in submit order
    drain = pdist.HostDrain(depth=2)
    t0 = drain.submit(pdist.all_gather_packed(rows, count, frame_offset=rank * B))
    t1 = drain.submit(pdist.all_gather_packed(rows * 0, count * 0, frame_offset=rank * B))
    assert np.array_equal(drain.collect(t0), flat) and drain.collect(t1).shape == (0, 14)
    t = pdist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    pdist.barrier()
    q.put((rank, g_counts.tolist(), flat[:, :2].tolist(), t))
    torch.distributed.destroy_process_group()


def test_all_gather_detections_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_rows = [[0.0, 0.0], [1.0, 1.0], [4.0, 10.0], [5.0, 11.0], [6.0, 12.0]]  # image idx shifted by rank * B
    for rank, counts, rows, tmax in res:
        assert counts == [2, 3]
        assert rows == want_rows      # every rank holds all detections, in global frame order
        assert tmax == 2.0            # MAX over ranks (bench.py timing contract)


def test_host_drain_refuses_to_overwrite_an_uncollected_batch():
    import pytest
    rows = torch.zeros((4, 14)); rows[0, 1] = 5
    packed = pdist.all_gather_packed(rows, torch.tensor([1], dtype=torch.int32))
    drain = pdist.HostDrain(depth=2)
    t0, t1 = drain.submit(packed), drain.submit(packed)
    with pytest.raises(RuntimeError):
        drain.submit(packed)
    assert drain.collect(t0).shape == (1, 14)
    with pytest.raises(RuntimeError):
        drain.collect(t0)
    t2 = drain.submit(packed)
    assert drain.collect(t1).shape == (1, 14) and drain.collect(t2).shape == (1, 14)


def test_all_gather_detections_world1_is_local():
    rows = torch.zeros((4, 14)); rows[0, 1] = 5
    g, c = pdist.all_gather_detections(rows, torch.tensor([1], dtype=torch.int32))
    assert g.shape == (1, 4, 14) and c.tolist() == [1]
    assert pdist.flatten_gathered(g, c).shape == (1, 14)


def test_upload_slots_are_adjacent_and_stack_as_a_view():
    """pipeline.alloc_adjacent carves a batch's blobs out of one allocation so that the colour and the
    depth blob of an RGB-D batch are back to back; stacked_view then hands both towers to the grouped
    trunk as ONE [2B,H,W,3] view (no concatenation), and refuses anything that is not adjacent."""
    import torch
    from posecnn_amd import pipeline
    a = torch.arange(4 * 6 * 8 * 3, dtype=torch.float32).reshape(4, 6, 8, 3)
    b = -a
    lab = torch.zeros(4, 6, 8, dtype=torch.int32)
    slot = pipeline.alloc_adjacent((a, b, None, lab), "cpu")
    assert slot[2] is None and [t.shape for t in slot if t is not None] == [a.shape, b.shape, lab.shape]
    assert slot[3].dtype == torch.int32 and all((t.data_ptr() - slot[0].data_ptr()) % 256 == 0 for t in slot if t is not None)   # (256-byte steps from the base; a CPU base is only 64-byte aligned)
    slot[0].copy_(a); slot[1].copy_(b)
    v = pipeline.stacked_view(slot[0], slot[1])
    assert v is not None and v.shape == (8, 6, 8, 3) and v.data_ptr() == slot[0].data_ptr()
    assert torch.equal(v[:4], a) and torch.equal(v[4:], b)
    assert pipeline.stacked_view(a, b) is None                      # separate allocations
    assert pipeline.stacked_view(slot[1], slot[0]) is None          # wrong order
    assert pipeline.stacked_view(slot[0], slot[1][:, :3]) is None   # shape mismatch
    odd = pipeline.alloc_adjacent((torch.zeros(3, 5, 7, 3), torch.zeros(3, 5, 7, 3)), "cpu")   # 1260 B: padded to 256
    assert pipeline.stacked_view(odd[0], odd[1]) is None


def test_listener_host_helpers():
    """posecnn_amd.listener: the depth-message decoding of ros/listener.py:42-51 and imdb.labels_to_image (lov.py:348-364)."""
    from posecnn_amd import listener
    d32 = np.array([[0.5, 1.2345], [0.0, 6.5]], np.float32)
    assert listener.depth_from_message(d32, "32FC1").tolist() == [[500, 1234], [0, 6500]]
    d16 = np.array([[1, 2], [3, 65535]], np.uint16)
    assert np.array_equal(listener.depth_from_message(d16, "16UC1"), d16) and listener.depth_from_message(d16, "8UC1") is None
    labels = np.array([[0, 1, 21], [2, 22, 5]], np.int32)
    img = listener.labels_to_image(labels)
    assert img.dtype == np.uint8 and img.shape == (2, 3, 3)
    assert img[0, 0].tolist() == [255, 255, 255] and img[0, 1].tolist() == [255, 0, 0] and img[0, 2].tolist() == [0, 0, 192]
    assert img[1, 0].tolist() == [0, 255, 0] and img[1, 1].tolist() == [0, 0, 0] and img[1, 2].tolist() == [255, 0, 255]   # class 22: no colour -> black
