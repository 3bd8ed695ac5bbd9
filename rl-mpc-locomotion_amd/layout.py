"""Flat record layout of one ``compute_contact_forces`` call.

The 13 positional arguments of the reference's plugin boundary
(``mpc_osqp.cc:578-591``, call site ``ConvexMPCLocomotion.py:171-185``) concatenated in call order
into one float32 row of length ``56 + 4*h``.  The same offsets are hard-coded in
``include/mpc_batch.h`` and the HIP kernels.
"""
IN_WEIGHTS = 0      # qp_weights[13]
IN_COM_POS = 13     # com_position[3]
IN_COM_VEL = 16     # com_velocity[3]
IN_RPY = 19         # com_roll_pitch_yaw[3]
IN_NORMAL = 22      # ground_normal_vec[3]
IN_ANGVEL = 25      # com_angular_velocity[3]
IN_CONTACT = 28     # foot_contact_states[4h]  row-major [step][leg]


def in_footpos(h): return 28 + 4 * h      # foot_positions_body_frame[12]  [leg][xyz]
def in_friction(h): return 40 + 4 * h     # foot_friction_coeffs[4]
def in_des_pos(h): return 44 + 4 * h      # desired_com_position[3]
def in_des_vel(h): return 47 + 4 * h      # desired_com_velocity[3]
def in_des_rpy(h): return 50 + 4 * h      # desired_com_roll_pitch_yaw[3]
def in_des_angvel(h): return 53 + 4 * h   # desired_com_angular_velocity[3]
def in_len(h): return 56 + 4 * h


def pack_args(h, qp_weights, com_position, com_velocity, com_roll_pitch_yaw, ground_normal_vec,
              com_angular_velocity, foot_contact_states, foot_positions_body_frame, foot_friction_coeffs,
              desired_com_position, desired_com_velocity, desired_com_roll_pitch_yaw,
              desired_com_angular_velocity, out):
    """Write the 13 reference arguments into ``out[56+4h]`` (any float dtype)."""
    import numpy as np
    def put(off, v, k):
        a = np.asarray(v, dtype=np.float64).reshape(-1)
        if a.size != k:
            raise ValueError(f"argument at offset {off}: expected {k} values, got {a.size}")
        out[off:off + k] = a
    put(IN_WEIGHTS, qp_weights, 13)
    put(IN_COM_POS, com_position, 3)
    put(IN_COM_VEL, com_velocity, 3)
    put(IN_RPY, com_roll_pitch_yaw, 3)
    put(IN_NORMAL, ground_normal_vec, 3)
    put(IN_ANGVEL, com_angular_velocity, 3)
    put(IN_CONTACT, foot_contact_states, 4 * h)
    put(in_footpos(h), foot_positions_body_frame, 12)
    put(in_friction(h), foot_friction_coeffs, 4)
    put(in_des_pos(h), desired_com_position, 3)
    put(in_des_vel(h), desired_com_velocity, 3)
    put(in_des_rpy(h), desired_com_roll_pitch_yaw, 3)
    put(in_des_angvel(h), desired_com_angular_velocity, 3)
    return out


def unpack_args(h, rec):
    """Inverse of pack_args: the 13 positional arguments of compute_contact_forces as float64 arrays."""
    import numpy as np
    r = np.asarray(rec, dtype=np.float64).reshape(-1)
    cut = lambda off, k: r[off:off + k].copy()
    return (cut(IN_WEIGHTS, 13), cut(IN_COM_POS, 3), cut(IN_COM_VEL, 3), cut(IN_RPY, 3), cut(IN_NORMAL, 3), cut(IN_ANGVEL, 3),
            cut(IN_CONTACT, 4 * h), cut(in_footpos(h), 12), cut(in_friction(h), 4), cut(in_des_pos(h), 3), cut(in_des_vel(h), 3),
            cut(in_des_rpy(h), 3), cut(in_des_angvel(h), 3))
