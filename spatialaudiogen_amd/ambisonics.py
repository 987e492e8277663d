"""First-order ambisonic geometry on the host: the speaker mesh and the real spherical-harmonic matrix the power-map
kernels project onto (reference: pyutils/ambisonics/distance.py:9-13 mesh, common.py:151-178 harmonics, decoder.py:9-28
'projection' decoding).  ACN channel order W,Y,Z,X with SN3D normalisation, as everywhere in the path."""
import numpy as np


def spherical_mesh(angular_res):
    """(phi, nu) grids in radians: azimuth from +180 down to -180 (exclusive) in steps of `angular_res` degrees along the
    columns, elevation from -90 to +90 along the rows (distance.py:9-13)."""
    phi = np.flip(np.arange(-180., 180., angular_res)) / 180. * np.pi
    nu = np.arange(-90., 90.1, angular_res) / 180. * np.pi
    return np.meshgrid(phi, nu)


def sh_order1(phi, nu):
    """Order-1 real harmonics, ACN/SN3D: [1, cos(nu) sin(phi), sin(nu), cos(nu) cos(phi)] (common.py:151-157)."""
    phi, nu = np.asarray(phi, np.float64), np.asarray(nu, np.float64)
    return np.stack([np.ones_like(phi), np.cos(nu) * np.sin(phi), np.sin(nu), np.cos(nu) * np.cos(phi)], -1)


def sh_matrix(angular_res):
    """[P, 4] projection matrix of the mesh, row-major over (elevation row, azimuth column)."""
    phi, nu = spherical_mesh(angular_res)
    return sh_order1(phi.reshape(-1), nu.reshape(-1))


def mesh_shape(angular_res):
    phi, _ = spherical_mesh(angular_res)
    return phi.shape
