"""Datasets / on-disk formats: every dataset class of the reference, constructed through the
reference's import paths, against the first-sample vectors tools/gen_golden.py --datasets took from the reference
(file-based ones on the synthetic files stored in the fixture)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def z():
    return np.load(os.path.join(GOLDEN, "datasets.npz"))


def check(z, tag, ds, attrs=(), rtol=0.0):
    x, f = ds[min(1, len(ds) - 1)]
    assert x.dtype == torch.float32 and f.dtype == torch.float32
    assert len(ds) == int(z[tag + "/len"])
    np.testing.assert_allclose(x.numpy(), z[tag + "/inputs"], rtol=rtol, atol=0)
    np.testing.assert_allclose(f.numpy(), z[tag + "/forcing"], rtol=rtol, atol=0)
    for a in attrs:
        v = getattr(ds, a)
        v = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        np.testing.assert_allclose(v, z[tag + "/attr_" + a], rtol=rtol, atol=0)


RECTS = ["Rectangle", "RectangleManufactured", "AdvDiff1dRectangle", "AdvDiff2dRectangle", "AllenCahnIceMeltRectangle",
         "RectangleManufacturedNonZeroBC", "RectangleHelmholtzManufactured", "RectangleHelmholtzDeltaForce",
         "RectangleManufacturedStokes", "RectangleIM", "RectangleIMBack"]


@pytest.mark.parametrize("name", RECTS)
def test_rectangle_datasets_bitwise(z, name):
    from DiffNet.datasets.single_instances import rectangles
    attrs = {"AllenCahnIceMeltRectangle": ("u0", "initial_guess"), "RectangleManufacturedNonZeroBC": ("u_exact",)}.get(name, ())
    check(z, "rect/" + name, getattr(rectangles, name)(domain_size=72), attrs)


def test_space_time_rectangle_consumes_the_same_random_numbers(z):
    from DiffNet.datasets.single_instances.rectangles import SpaceTimeRectangleManufactured
    np.random.seed(1234)
    torch.manual_seed(1234)
    check(z, "rect/SpaceTimeRectangleManufactured", SpaceTimeRectangleManufactured(domain_size=24), ("u0", "initial_guess"))


def test_cuboid_circle_lshaped_bitwise(z):
    from DiffNet.datasets.single_instances.cuboids import Cuboid, CuboidManufactured
    from DiffNet.datasets.single_instances.circles import CircleIMBack
    from DiffNet.datasets.single_instances.Lshaped import LShaped
    check(z, "cuboid/Cuboid", Cuboid(domain_size=12))
    check(z, "cuboid/CuboidManufactured", CuboidManufactured(domain_size=12))
    check(z, "circle/CircleIMBack", CircleIMBack(domain_size=72))
    check(z, "lshaped/LShaped", LShaped(domain_size=72))


def test_image_datasets_from_files(z, tmp_path):
    from DiffNet.datasets.single_instances import images as s_images
    from DiffNet.datasets.parametric import images as p_images
    d = tmp_path / "imgs"
    d.mkdir()
    for k in range(3):
        (d / f"shape{k}.png").write_bytes(z[f"files/img{k}"].tobytes())
    check(z, "img/single_ImageIMBack", s_images.ImageIMBack(str(d / "shape1.png")))
    check(z, "img/single_Disk", s_images.Disk(str(d / "shape2.png")))
    for name in ("ImageIMBack", "ImageIMBackObject", "ImageIMBackNeumann"):
        check(z, "img/param_" + name, getattr(p_images, name)(str(d)))
    (d / "notes.txt").write_text("x")
    with pytest.raises(ValueError):
        p_images.ImageIMBack(str(d))                    # the reference rejects unknown extensions the same way


def test_voxel_raw_format(z, tmp_path):
    from DiffNet.datasets.single_instances.voxels import VoxelIMBackRAW, load_raw
    prefix = str(tmp_path / "obj_")
    open(prefix + "inouts.raw", "wb").write(z["files/vox_raw"].tobytes())
    open(prefix + "VoxelConfig.txt", "wb").write(z["files/vox_cfg"].tobytes())
    vox, numdiv, grid, bmin = load_raw(prefix)
    assert vox.shape == (9, 7, 5) and list(numdiv) == [9, 7, 5] and np.allclose(grid, 0.1) and np.allclose(bmin, 0.0)
    check(z, "vox/VoxelIMBackRAW", VoxelIMBackRAW(prefix, domain_size=48))


def test_kl_sum_fields(z, tmp_path):
    from DiffNet.datasets.parametric import klsum as p_klsum
    from DiffNet.datasets.single_instances import klsum as s_klsum
    from DiffNet import gen_input_calc
    coeffs = z["files/kl_coeffs"]
    np.save(tmp_path / "coeffs.npy", coeffs)
    np.savetxt(tmp_path / "coeff.txt", coeffs[3])
    for eta in (0.1, 0.2, 0.5, 0.7, 1.0):          # computed roots reproduce the reference's tabulated digits
        np.testing.assert_allclose(gen_input_calc.calculate_omega_based_on_eta(eta), z[f"kl/omega_{eta}"], rtol=0, atol=2e-13)
    np.testing.assert_allclose(gen_input_calc.generate_diffusivity_tensor(coeffs[0], output_size=6, nsd=3), z["kl/nu3d"], rtol=1e-12)
    ks = p_klsum.KLSumStochastic(str(tmp_path / "coeffs.npy"), domain_size=20, kl_terms=6)
    check(z, "kl/param_KLSumStochastic", ks, rtol=2e-7)          # block-evaluated in float64, stored float32: <= 1 float32 ulp
    np.testing.assert_allclose(np.stack([ks[i][0].numpy() for i in range(len(ks))]), z["kl/param_all"], rtol=2e-7)
    ks4 = p_klsum.KLSumStochastic(str(tmp_path / "coeffs.npy"), domain_size=20, kl_terms=4)
    np.testing.assert_allclose(np.stack([ks4[i][0].numpy() for i in range(len(ks4))]), z["kl/param_terms4"], rtol=2e-7)
    check(z, "kl/param_Dataset", p_klsum.Dataset(str(tmp_path / "coeff.txt"), domain_size=20), rtol=1e-7)
    check(z, "kl/single_Dataset", s_klsum.Dataset(str(tmp_path / "coeff.txt"), domain_size=20), rtol=1e-7)
    with pytest.raises(FileNotFoundError):
        s_klsum.Dataset(str(tmp_path / "missing.txt"))


def test_device_loader_batches_on_cpu():
    from diffnet_amd.datasets import DeviceLoader
    from DiffNet.datasets.single_instances.rectangles import RectangleManufactured
    dl = DeviceLoader(RectangleManufactured(16), batch_size=8, device="cpu", max_samples=20)
    batches = list(dl)
    assert len(dl) == 3 and [b[0].shape[0] for b in batches] == [8, 8, 4]
    assert batches[0][0].shape == (8, 3, 16, 16) and batches[0][1].shape == (8, 1, 16, 16)
    dl2 = DeviceLoader(RectangleManufactured(16), batch_size=8, device="cpu", max_samples=20, shuffle=True, drop_last=True)
    assert len(list(dl2)) == 2


def test_point_cloud_dataset_of_the_flagship_script(z, tmp_path):
    """`PointClouds` (IBN_2D.py:35-84): npz layout, affine placement, squared-segment 'area' weights, sample layout."""
    from DiffNet.datasets.parametric.pointclouds import PointClouds
    prefix = str(tmp_path) + os.sep
    np.savez(prefix + "point_cloud.npz", z["files/pc_points"])
    np.savez(prefix + "normals.npz", z["files/pc_normals"])
    ds = PointClouds(prefix, type='val', domain_size=24)
    x, f, snk = ds[4]
    assert len(ds) == int(z["pc/len"]) and x.shape == (40, 5)
    np.testing.assert_array_equal(x.numpy(), z["pc/inputs"])
    np.testing.assert_array_equal(f.numpy(), z["pc/forcing"])
    np.testing.assert_array_equal(snk.numpy(), z["pc/sink"])
    np.testing.assert_array_equal(ds.area, z["pc/area"])


def test_topo3d_dataset_layout(tmp_path):
    """`TopoDataset3D` (IBN_3D.py:76-106): one npz per object, (source, sink, forcing) samples, 100 / 25 split."""
    from DiffNet.datasets.parametric.topo3d import TopoDataset3D, write_blob_objects
    write_blob_objects(str(tmp_path), n_objects=5, domain_size=8, seed=1)
    ds = TopoDataset3D(str(tmp_path), domain_size=8, mode='train')
    assert len(ds) == 5 and len(TopoDataset3D(str(tmp_path), domain_size=8, mode='val')) == 0
    src, snk, f = ds[2]
    assert src.shape == snk.shape == f.shape == (1, 8, 8, 8) and float(f.abs().max()) == 0.0
    ref = np.load(os.path.join(str(tmp_path), ds.list_IDs[2]))['arr_0']
    np.testing.assert_array_equal(src.numpy(), ref)
    b = snk[0].numpy()
    assert b[0].all() and b[-1].all() and b[:, 0].all() and b[:, :, -1].all() and b[1:-1, 1:-1, 1:-1].sum() == 0
