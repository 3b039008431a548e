"""Constants of the demo/test configuration of the reference (data, not code).

Sources: experiments/cfgs/lov_color_2d.yml, lib/fcn/config.py:67,216,242, tools/demo.py:100-101,
lib/datasets/lov.py:27-38, data/LOV/extents.txt, lib/networks/vgg16_convs.py:20-29.
"""
from dataclasses import dataclass, field

import numpy as np

# lib/datasets/lov.py:27-30 — 21 YCB objects + background
LOV_CLASSES = (
    "__background__", "002_master_chef_can", "003_cracker_box", "004_sugar_box",
    "005_tomato_soup_can", "006_mustard_bottle", "007_tuna_fish_can", "008_pudding_box",
    "009_gelatin_box", "010_potted_meat_can", "011_banana", "019_pitcher_base",
    "021_bleach_cleanser", "024_bowl", "025_mug", "035_power_drill", "036_wood_block",
    "037_scissors", "040_large_marker", "051_large_clamp", "052_extra_large_clamp",
    "061_foam_brick")

# lib/datasets/lov.py:38
LOV_SYMMETRY = np.array([0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1], dtype=np.float32)

# data/LOV/extents.txt (metres), row 0 = background (lov.py:163-170 loads rows 1.. from the file)
LOV_EXTENTS = np.array([
    [0.0, 0.0, 0.0],
    [0.105098, 0.103336, 0.147140], [0.072948, 0.167432, 0.223122], [0.051228, 0.097062, 0.184740],
    [0.068346, 0.070898, 0.118506], [0.099712, 0.071530, 0.215002], [0.085656, 0.085848, 0.041788],
    [0.140458, 0.136312, 0.044982], [0.092226, 0.102030, 0.037278], [0.106770, 0.061462, 0.099400],
    [0.146328, 0.202874, 0.039542], [0.159810, 0.157306, 0.293620], [0.112422, 0.072590, 0.277178],
    [0.161696, 0.163252, 0.060978], [0.133400, 0.094318, 0.084588], [0.202122, 0.229442, 0.061552],
    [0.106668, 0.108480, 0.240242], [0.110210, 0.257878, 0.015808], [0.021110, 0.125212, 0.019532],
    [0.140818, 0.174792, 0.040068], [0.210450, 0.185262, 0.036514], [0.052900, 0.077960, 0.067918],
], dtype=np.float32)

# data/LINEMOD/extents.txt (metres; 15 objects, lib/datasets/linemod.py:35-37), row 0 = background.
# BASELINE config 4 ("LINEMOD 13-class") uses the first 13 objects: C = 14.
LINEMOD_EXTENTS_ALL = np.array([
    [0.0, 0.0, 0.0],
    [0.075868, 0.077600, 0.091770], [0.215670, 0.121856, 0.219410], [0.166432, 0.165318, 0.074472],
    [0.136660, 0.143030, 0.100498], [0.100792, 0.181796, 0.193734], [0.067010, 0.127632, 0.117456],
    [0.117580, 0.091512, 0.094622], [0.229476, 0.075472, 0.208002], [0.104430, 0.077408, 0.085698],
    [0.150184, 0.107076, 0.069242], [0.036722, 0.077866, 0.172816], [0.100888, 0.108496, 0.090800],
    [0.258226, 0.118482, 0.141132], [0.203146, 0.117752, 0.213116], [0.093918, 0.147434, 0.184748],
], dtype=np.float32)
LINEMOD_EXTENTS = LINEMOD_EXTENTS_ALL[:14].copy()
# lib/datasets/linemod.py:45 (eggbox symmetric), cut to the same 13 objects
LINEMOD_SYMMETRY = np.array([0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0], dtype=np.float32)

# tools/demo.py:100 (YCB-Video camera) and :101
DEMO_INTRINSICS = np.array([[1066.778, 0.0, 312.9869], [0.0, 1067.487, 241.3109], [0.0, 0.0, 1.0]], dtype=np.float64)
DEMO_FACTOR_DEPTH = 10000.0

# lib/fcn/config.py:242 (BGR order). float64 like the reference's: `im_orig -= cfg.PIXEL_MEANS` on a float32 image is then
# numpy's float32 -= float64, i.e. float32(double(x) - mean) — not x - float32(mean), which differs by up to 2.5e-6
PIXEL_MEANS = np.array([[[102.9801, 115.9465, 122.7717]]], dtype=np.float64)

NUM_MODEL_POINTS = 2620  # min over data/LOV/models/*/points.xyz (lov.py:141-158)


@dataclass
class TestConfig:
    """The knobs `vgg16_convs` is built with for tools/demo.py / tools/test_net.py."""
    __test__ = False  # not a pytest class
    input_format: str = "COLOR"          # lov_color_2d.yml:2
    num_classes: int = 22                # lov_color_2d.yml:14
    num_units: int = 64                  # lov_color_2d.yml:15
    scales_base: tuple = (1.0,)          # lov_color_2d.yml:39
    threshold_label: float = 1.0         # lov_color_2d.yml:26
    vote_threshold: float = -1.0         # config.py:67,216; tools/test_net.py:96
    vote_percentage: float = 0.02        # vgg16_convs.py:24,29
    skip_pixels: int = 10                # vgg16_convs.py:22,27
    vertex_reg_2d: bool = True
    pose_reg: bool = True
    is_train: bool = False
    margin: float = 0.01                 # vgg16_convs.py:200
    extents: np.ndarray = field(default_factory=lambda: LOV_EXTENTS.copy())
    symmetry: np.ndarray = field(default_factory=lambda: LOV_SYMMETRY.copy())


def make_meta_data(K, im_scale=1.0, voxel_step=(0.0, 0.0, 0.0), voxel_min=(0.0, 0.0, 0.0),
                   pose_world2live=None, pose_live2world=None):
    """meta_data[48] exactly as lib/fcn/test.py:130-149 builds it (K*scale, K[2,2]=1, pinv(K),
    voxel step/min; the two 3x4 poses stay zero there — optional here for the backproject op)."""
    K = np.array(K, dtype=np.float64) * im_scale
    K[2, 2] = 1
    Kinv = np.linalg.pinv(K)
    m = np.zeros(48, dtype=np.float32)
    m[0:9] = K.flatten()
    m[9:18] = Kinv.flatten()
    if pose_world2live is not None:
        m[18:30] = np.asarray(pose_world2live, dtype=np.float32).flatten()
    if pose_live2world is not None:
        m[30:42] = np.asarray(pose_live2world, dtype=np.float32).flatten()
    m[42:45] = voxel_step
    m[45:48] = voxel_min
    return m
