"""Drop-in for the reference module `animate` (animate.py:6-140): the NACA-00xx outline flown along a trajectory.

    from animate import Airfoil
    Airfoil(20, xx_star, xx_ref, xlim=[0, 17], ylim=[-5, 5]).run_animation(name='step')   # main_newton_method.py:226-227

Host-side post-processing, no device work: the outline of every frame is one broadcast over the
trajectory (`poses`), the figure is the reference's (same axes, artists, labels, inset, 15 fps GIF at
`Figures/AircraftBehavior_<name>.gif`).  Matplotlib is imported when a figure is asked for, not at import, and under a
headless backend `plt.show()` returns at once.  Kept quirks: the horizon is one second whatever the trajectory
(animate.py:45-47, `TT = int(1.0/dt)` frames, so a trajectory with another number of columns fails in the inset plot as it
does there); the reference outline is drawn at the LAST reference state in every frame (animate.py:108); x is mirrored
twice (−x_loc in the pose, −v[0] when drawn, animate.py:31, :105)."""
import os

import numpy as np

# half thickness of a NACA 00xx section of unit chord, terms in √x, x, x², x³, x⁴ (open trailing edge)
_NACA_TERMS = ((0.2969, 0.5), (-0.1260, 1), (-0.3516, 2), (0.2843, 3), (-0.1015, 4))


def naca_outline(th, n=100):
    """-> (3, 2n) homogeneous points: upper surface from nose to tail, then the lower one (animate.py:116-140).
    `th` is the thickness in per cent of the chord."""
    x = np.linspace(0, 1, n)
    half = np.zeros_like(x)
    for coef, power in _NACA_TERMS:                      # summed in the order the reference writes the terms
        half = half + coef * (np.sqrt(x) if power == 0.5 else x ** power)
    half = (5 * (th / 100.0)) * half
    return np.stack([np.concatenate([x, x]), np.concatenate([half, -half]), np.ones(2 * n)])


class Airfoil:
    """Airfoil(th, xx_star, xx_ref, dt=1e-3, xlim=[0, 15], ylim=[-4, 4])   (animate.py:6-18)"""

    def __init__(self, th, xx_star, xx_ref, dt=1e-3, xlim=[0, 15], ylim=[-4, 4]):
        self.th = th
        self.xlim = xlim
        self.ylim = ylim
        self.airfoil = naca_outline(th)
        self.xx_star = xx_star
        self.xx_ref = xx_ref
        self.dt = dt

    # ---- geometry -------------------------------------------------------------------------------------------------
    def poses(self, theta, x_loc, y_loc):
        """Outline at F poses at once -> (F, 3, 2n): rotation by theta about z, translation (−x_loc, y_loc)."""
        theta, x_loc, y_loc = (np.atleast_1d(np.asarray(a, dtype=np.float64)) for a in (theta, x_loc, y_loc))
        c, s = np.cos(theta)[:, None], np.sin(theta)[:, None]
        px, py, one = self.airfoil
        out = np.empty((theta.shape[0], 3, px.shape[0]))
        out[:, 0] = c * px - s * py - x_loc[:, None] * one
        out[:, 1] = s * px + c * py + y_loc[:, None] * one
        out[:, 2] = one
        return out

    def update_pose(self, theta, x_loc, y_loc):
        """-> (3, 2n) outline at one pose (animate.py:20-36)."""
        return self.poses(theta, x_loc, y_loc)[0]

    # ---- figure ---------------------------------------------------------------------------------------------------
    def run_animation(self, name='', stride=1, fps=15, out_dir='Figures'):
        """Writes `<out_dir>/AircraftBehavior_<name>.gif` and shows the figure (animate.py:38-83).  `stride` > 1 keeps
        every stride-th frame (the reference renders all 1/dt of them), `fps` and `out_dir` are the reference's constants.
        Returns the path of the GIF."""
        import matplotlib.pyplot as plt
        import matplotlib.animation as animation
        from matplotlib.ticker import MultipleLocator

        xx_star, xx_ref, dt = self.xx_star, self.xx_ref, self.dt
        tf = 1.0
        TT = int(tf / dt)
        time = np.linspace(0, tf, TT)
        # every frame's outline in one go; the reference outline does not move
        self._frames = self.poses(xx_star[3, :TT], xx_star[0, :TT], xx_star[1, :TT])
        self._ref_frame = self.update_pose(xx_ref[3, -1], xx_ref[0, -1], xx_ref[1, -1])

        fig = plt.figure()
        ax = fig.add_subplot(111, autoscale_on=False, xlim=tuple(self.xlim[:2]), ylim=tuple(self.ylim[:2]))
        ax.grid()
        ax.set_yticklabels([])
        ax.set_xticklabels([])
        self.line0, = ax.plot([], [], 'o-', lw=2, c='b', label='Optimal')
        self.line1, = ax.plot([], [], '*-', lw=2, c='g', dashes=[2, 2], label='Reference')
        self.time_template = 't = %.1f s'
        self.time_text = ax.text(0.05, 0.9, '', transform=ax.transAxes)
        fig.gca().set_aspect('equal', adjustable='box')

        inset = fig.add_axes([0.64, 0.13, 0.2, 0.2])          # altitude over time, lower right
        inset.xaxis.set_major_locator(MultipleLocator(2))
        inset.yaxis.set_major_locator(MultipleLocator(0.25))
        inset.set_xticklabels([])
        inset.grid(which='both')
        inset.plot(time, xx_star[1, :], c='b')
        inset.plot(time, xx_ref[1, :], color='g', dashes=[2, 1])
        self.point1, = inset.plot([], [], 'o', lw=2, c='b')

        which = range(0, TT, max(1, int(stride)))
        ani = animation.FuncAnimation(fig, self.animate, which, interval=1, blit=False, init_func=self.anime_init)
        os.makedirs(out_dir, exist_ok=True)                   # the reference relies on its checked-in Figures/ directory
        path = os.path.join(out_dir, 'AircraftBehavior_%s.gif' % name)
        ani.save(path, writer=animation.PillowWriter(fps=fps))
        ax.legend(loc="lower left")
        plt.show()
        return path

    def anime_init(self):
        """Empty artists (animate.py:85-96)."""
        for artist in (self.line0, self.line1, self.point1):
            artist.set_data([], [])
        self.time_text.set_text('')
        return self.line0, self.line1, self.time_text, self.point1

    def animate(self, i):
        """Frame i (animate.py:99-114)."""
        frames = getattr(self, "_frames", None)
        v = frames[i] if frames is not None and i < len(frames) else \
            self.update_pose(self.xx_star[3, i], self.xx_star[0, i], self.xx_star[1, i])
        vr = getattr(self, "_ref_frame", None)
        if vr is None:
            vr = self.update_pose(self.xx_ref[3, -1], self.xx_ref[0, -1], self.xx_ref[1, -1])
        self.line0.set_data(-v[0, :], v[1, :])
        self.line1.set_data(-vr[0, :], vr[1, :])
        self.point1.set_data([i * self.dt], [self.xx_star[1, i]])
        self.time_text.set_text(self.time_template % (i * self.dt))
        return self.line0, self.line1, self.time_text, self.point1
