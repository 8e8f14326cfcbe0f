"""Episode animation for the list-API env: the role of `env.enable_animation()` / `env.save_animation(path)` in the reference's
example.py:59,67-70 (there POGEMA's AnimationMonitor, a third-party wrapper that is not part of /root/reference).

Own writer, host side only, off the hot path: GridEnv records the positions it hands out anyway (one [n, 2] int16 array per step) and
`write_svg` turns them into one self-contained SVG -- obstacles as squares, goals as rings, agents as discs whose centres move by SMIL
`<animate>` key frames (one per env step), an agent and its goal in the same colour.  The 5-cell obstacle border of the padded grid is cropped.
"""
import colorsys

import numpy as np

CELL = 20          # pixels per cell
BORDER = 5         # cells of padding on every side of the padded grid (maps.pad)


def _colour(i, n):
    r, g, b = colorsys.hsv_to_rgb((i * 0.61803398875) % 1.0, 0.65, 0.85)      # golden-ratio hues: neighbours in id are far apart in colour
    return f"#{int(r * 255):02x}{int(g * 255):02x}{int(b * 255):02x}"


def write_svg(path, grid_padded, positions, goals, seconds_per_step=0.25):
    """grid_padded: uint8 [H, W] (non-zero = obstacle); positions: sequence of [n, 2] (row, col) arrays in padded coordinates, one per frame
    (reset + every step); goals: [n, 2] or one array per frame (lifelong episodes: the goal ring jumps when it changes)."""
    grid = np.asarray(grid_padded)[BORDER:-BORDER, BORDER:-BORDER]
    H, W = grid.shape
    frames = [np.asarray(p, dtype=np.int64) - BORDER for p in positions]
    if not frames:
        raise ValueError("nothing recorded: enable_animation() before reset()")
    n, F = frames[0].shape[0], len(frames)
    goals = np.asarray(goals, dtype=np.int64)
    goal_frames = [goals - BORDER] * F if goals.ndim == 2 else [np.asarray(g, dtype=np.int64) - BORDER for g in goals]
    dur = max(F - 1, 1) * seconds_per_step
    key_times = ";".join(f"{(f / max(F - 1, 1)):.5f}" for f in range(F)) if F > 1 else "0"

    def centre(rc):
        return (rc[1] + 0.5) * CELL, (rc[0] + 0.5) * CELL      # x from the column, y from the row

    out = [f'<svg xmlns="http://www.w3.org/2000/svg" viewBox="0 0 {W * CELL} {H * CELL}" width="{W * CELL}" height="{H * CELL}">',
           f'<rect width="{W * CELL}" height="{H * CELL}" fill="#ffffff"/>']
    for r, c in np.argwhere(grid != 0):
        out.append(f'<rect x="{c * CELL}" y="{r * CELL}" width="{CELL}" height="{CELL}" fill="#84a1ae"/>')

    def animated(attr_x, attr_y, pts):
        if F == 1 or all(p == pts[0] for p in pts):
            return ""
        xs = ";".join(f"{p[0]:g}" for p in pts)
        ys = ";".join(f"{p[1]:g}" for p in pts)
        common = f'dur="{dur:g}s" keyTimes="{key_times}" repeatCount="indefinite"'
        return f'<animate attributeName="{attr_x}" values="{xs}" {common}/><animate attributeName="{attr_y}" values="{ys}" {common}/>'

    for a in range(n):
        col = _colour(a, n)
        gp = [centre(gf[a]) for gf in goal_frames]
        out.append(f'<circle cx="{gp[0][0]:g}" cy="{gp[0][1]:g}" r="{CELL * 0.38:g}" fill="none" stroke="{col}" stroke-width="2">'
                   + animated("cx", "cy", gp).replace("<animate ", '<animate calcMode="discrete" ') + "</circle>")
    for a in range(n):
        col = _colour(a, n)
        pp = [centre(fr[a]) for fr in frames]
        out.append(f'<circle cx="{pp[0][0]:g}" cy="{pp[0][1]:g}" r="{CELL * 0.3:g}" fill="{col}">' + animated("cx", "cy", pp) + "</circle>")
    out.append("</svg>")
    import os
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return path
