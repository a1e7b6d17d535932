from diffnet_amd.cuboid_mesh import CuboidMesh  # noqa: F401
