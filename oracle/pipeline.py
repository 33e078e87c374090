"""numpy restatement of the reference's loader-side pipeline with the random draws as INPUTS.

TEST INFRASTRUCTURE ONLY (see oracle/ops.py).  Reference: datasets/ShapeNet55Dataset.py:67-119 (pc_norm,
random_sample, __getitem__) and datasets/corrupt_util.py -- _pc_normalize :7-17, corrupt_scale_nonorm_2p :82-92,
corrupt_tranlate :130-140, corrupt_jitter :179-191, corrupt_rotate_360 :241-263, corrupt_reflection :390-409,
corrupt_shear_p5 :412-428, corrupt_add_global :830-841, corrupt_add_local :844-870, density :875-897, the dispatcher
corrupt_data :1046-1096 and augment_data :1155-1175.

Every function below is the reference's arithmetic with what it DRAWS turned into an argument (the fixture generator
records the draws of the live functions and checks that these restatements then reproduce their outputs).  dtype
transitions follow the reference: translate / scale return float32, the matrix maps float64 (numpy upcasts a float32
cloud against a float64 matrix), __getitem__ ends with `.float()`.
"""
import numpy as np


def pc_normalize(pc):
    """corrupt_util._pc_normalize :7-17 (the 'norm' augmentation; same dtype as the input)."""
    centroid = np.mean(pc, axis=0)
    pc = pc - centroid
    m = np.max(np.sqrt(np.sum(pc ** 2, axis=1)))
    return pc / m


def affine_steps(pc, steps):
    """'affine_r3' (:1062-1070) after its draws: steps = [(kind, value)] in the order drawn; kind 'translate'
    (value xyz (3,), :139-140), 'scale_nonorm' (xyz (3,), :91-92), or 'matrix' (R (3,3): rotate :262-263, reflection
    :408-409, shear :425-428 -- all `np.dot(pointcloud, R)`)."""
    for kind, v in steps:
        if kind == 'translate':
            pc = (pc + v).astype('float32')
        elif kind == 'scale_nonorm':
            pc = np.multiply(pc, v).astype('float32')
        elif kind == 'matrix':
            pc = np.dot(pc, v)
        else:
            raise KeyError(kind)
    return pc


def rotation_matrix(angles):
    """corrupt_rotate_360 :252-262 from its three angles."""
    Rx = np.array([[1, 0, 0], [0, np.cos(angles[0]), -np.sin(angles[0])], [0, np.sin(angles[0]), np.cos(angles[0])]])
    Ry = np.array([[np.cos(angles[1]), 0, np.sin(angles[1])], [0, 1, 0], [-np.sin(angles[1]), 0, np.cos(angles[1])]])
    Rz = np.array([[np.cos(angles[2]), -np.sin(angles[2]), 0], [np.sin(angles[2]), np.cos(angles[2]), 0], [0, 0, 1]])
    return np.dot(Rz, np.dot(Ry, Rx))


def reflection_matrix(signs):
    """corrupt_reflection :397-408 from its three +-1 draws."""
    return np.diag(np.asarray(signs, dtype=np.float64))


def shear_matrix(shear):
    """corrupt_shear_p5 :423-427 from its six draws."""
    return np.array([[1, shear[0], shear[1]], [shear[2], 1, shear[3]], [shear[4], shear[5], 1]])


def jitter(pc, level, noise):
    """corrupt_jitter :179-191: sigma = 0.01 (level + 1); noise = np.random.randn(N, C)."""
    sigma = 0.01 * (level + 1)
    return pc + sigma * noise


def sphere_points(radius_u, costheta_u, phi_u):
    """_sample_points_inside_unit_sphere :42-56 from its three (n,1) uniform draws."""
    radius = np.power(radius_u, 1 / 3)
    theta = np.arccos(costheta_u)
    x = radius * np.sin(theta) * np.cos(phi_u)
    y = radius * np.sin(theta) * np.sin(phi_u)
    z = radius * np.cos(theta)
    return np.concatenate([x, y, z], axis=1)


def add_global(pc, level, radius_u, costheta_u, phi_u):
    """corrupt_add_global :830-841: int(P (level + 1) 0.1) points uniform in the unit ball, appended."""
    npoints = int(pc.shape[0] * (level + 1) * 0.1)
    extra = sphere_points(radius_u, costheta_u, phi_u)
    return np.concatenate([pc, extra[:npoints]], axis=0)


def add_local(pc, level, order, sizes, sigmas, noise):
    """corrupt_add_local :844-870: `order` = the permutation _shuffle_pointcloud applied (:854), sizes / sigmas per
    cluster, noise = the cluster-by-cluster np.random.randn draws stacked (total, 3).  -> shuffled cloud + clusters."""
    num_points = pc.shape[0]
    total = int(num_points * (level + 1) * 0.1)
    pc = pc[order]
    add_pcd = np.zeros_like(pc)
    num_added = 0
    for i, (K, sigma) in enumerate(zip(sizes, sigmas)):
        add_pcd[num_added:num_added + K, :] = np.copy(pc[i:i + 1, :])
        add_pcd[num_added:num_added + K, :] = add_pcd[num_added:num_added + K, :] + sigma * noise[num_added:num_added + K]
        num_added += K
    assert num_added == total
    dist = np.sum(add_pcd ** 2, axis=1, keepdims=True).repeat(3, axis=1)
    add_pcd[dist > 1] = add_pcd[dist > 1] / dist[dist > 1]
    return np.concatenate([pc, add_pcd], axis=0)[:num_points + total]


def density_keep(pc, level, v_raw, r_list):
    """density :875-897: v_raw = np.random.normal(0, 1, 3) before normalisation, r_list = uniform(0, 1, P).
    -> bool keep mask (the reference returns pc[mask])."""
    gate = level / 4.0 + 0.1
    v_point = v_raw / np.linalg.norm(v_raw)
    dist = np.sqrt((v_point ** 2).sum())
    max_dist, min_dist = dist + 1, dist - 1
    d = np.linalg.norm(pc - v_point.reshape(1, 3), axis=1)
    d = (d - min_dist) / (max_dist - min_dist)
    return d * gate < r_list


def random_sample(pc, num, permutation, refill=None):
    """ShapeNet.random_sample :76-88: permutation = the shuffled arange it draws; refill = the np.random.choice
    indices of the short-cloud branch."""
    if pc.shape[0] < num:
        pc = np.vstack((pc, pc[refill]))
    return pc[permutation[:num]]
