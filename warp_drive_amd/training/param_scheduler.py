"""Piecewise-linear hyper-parameter schedules
(reference warp_drive/training/utils/param_scheduler.py): a number, or a list of
[timestep, value] knots interpolated linearly and held constant outside the knots."""
import numpy as np


class ParamScheduler:
    def __init__(self, schedule):
        if isinstance(schedule, (int, float)):
            self.knots = None
            self.value = float(schedule)
        else:
            pts = sorted((float(t), float(v)) for t, v in schedule)
            assert len(pts) >= 1
            self.knots = (np.array([p[0] for p in pts]), np.array([p[1] for p in pts]))

    def get_param_value(self, timestep):
        if self.knots is None:
            return self.value
        return float(np.interp(float(timestep), self.knots[0], self.knots[1]))
