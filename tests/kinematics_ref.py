"""TEST HELPER: all-link kinematics in plain torch -- ``rigid_body_states`` (N, num_links, 13) in the layout of Isaac Gym's
``acquire_rigid_body_state_tensor`` (legged_robot.py:113,134: position 3, quaternion xyzw 4, linear velocity 3, angular
velocity 3, world frame, link-frame origins) from (root_states, dof_pos, dof_vel) and the robot model.

An independent third derivation next to the oracle's forward kinematics and the step kernel's own walk: the product publishes
GRX_T_RIGID_BODY_STATES from the fused kernel (round 3; rounds 1-2 shipped this module as envs/kinematics.py and evaluated it
lazily in eager torch)."""
import torch


def _quat_to_matrix(q):
    x, y, z, w = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1).reshape(q.shape[:-1] + (3, 3))


def _matrix_to_quat(R):
    """Rotation matrices (..., 3, 3) -> unit quaternions xyzw, branch-free (largest-component form)."""
    m = R
    t = torch.stack([1 + m[..., 0, 0] - m[..., 1, 1] - m[..., 2, 2],
                     1 - m[..., 0, 0] + m[..., 1, 1] - m[..., 2, 2],
                     1 - m[..., 0, 0] - m[..., 1, 1] + m[..., 2, 2],
                     1 + m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2]], dim=-1)          # 4 x^2, 4 y^2, 4 z^2, 4 w^2
    cand = torch.stack([
        torch.stack([t[..., 0], m[..., 0, 1] + m[..., 1, 0], m[..., 0, 2] + m[..., 2, 0], m[..., 2, 1] - m[..., 1, 2]], -1),
        torch.stack([m[..., 0, 1] + m[..., 1, 0], t[..., 1], m[..., 1, 2] + m[..., 2, 1], m[..., 0, 2] - m[..., 2, 0]], -1),
        torch.stack([m[..., 0, 2] + m[..., 2, 0], m[..., 1, 2] + m[..., 2, 1], t[..., 2], m[..., 1, 0] - m[..., 0, 1]], -1),
        torch.stack([m[..., 2, 1] - m[..., 1, 2], m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] - m[..., 0, 1], t[..., 3]], -1)], dim=-2)
    best = t.argmax(dim=-1)
    q = torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)
    q = q / q.norm(dim=-1, keepdim=True)
    return torch.where(q[..., 3:4] < 0, -q, q)


class BodyKinematics:
    def __init__(self, rm, device):
        import numpy as np
        f = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=device)
        self.parent = list(rm.parent)
        self.axis, self.rot0, self.jpos = f(rm.joint_axis), f(rm.joint_rot0), f(rm.joint_pos)
        self.link_body = torch.as_tensor(rm.link_body, dtype=torch.long, device=device)
        self.link_rot, self.link_pos = f(rm.link_rot), f(rm.link_pos)
        self.num_bodies, self.num_links = rm.num_bodies, rm.num_links

    def body_frames(self, root_states, q, qd):
        """world rotation (N, nb, 3, 3), origin (N, nb, 3), linear velocity of the origin, angular velocity"""
        N = root_states.shape[0]
        R = [None] * self.num_bodies
        p, v, w = [None] * self.num_bodies, [None] * self.num_bodies, [None] * self.num_bodies
        R[0] = _quat_to_matrix(root_states[:, 3:7])
        p[0], v[0], w[0] = root_states[:, 0:3], root_states[:, 7:10], root_states[:, 10:13]
        for b in range(1, self.num_bodies):
            pb = self.parent[b]
            a = self.axis[b]
            K = torch.tensor([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]], device=q.device, dtype=q.dtype)
            ang = q[:, b - 1]
            rot = (torch.eye(3, device=q.device) + torch.sin(ang)[:, None, None] * K + (1 - torch.cos(ang))[:, None, None] * (K @ K))
            Rj = R[pb] @ self.rot0[b]                       # joint frame in the world
            R[b] = Rj @ rot
            off = (R[pb] @ self.jpos[b])                    # parent origin -> joint origin, world
            p[b] = p[pb] + off
            aw = Rj @ a
            w[b] = w[pb] + aw * qd[:, b - 1:b]
            v[b] = v[pb] + torch.cross(w[pb], off, dim=-1)
        return torch.stack(R, 1), torch.stack(p, 1), torch.stack(v, 1), torch.stack(w, 1)

    def rigid_body_states(self, root_states, q, qd):
        R, p, v, w = self.body_frames(root_states, q, qd)
        Rb, pb, vb, wb = R[:, self.link_body], p[:, self.link_body], v[:, self.link_body], w[:, self.link_body]
        off = (Rb @ self.link_pos[None, :, :, None]).squeeze(-1)
        Rl = Rb @ self.link_rot[None]
        return torch.cat([pb + off, _matrix_to_quat(Rl), vb + torch.cross(wb, off, dim=-1), wb], dim=-1)
