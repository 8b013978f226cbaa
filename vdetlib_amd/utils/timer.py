"""tic/toc timer with the attributes T-CNN scripts read (reference utils/timer.py:10-32): ``total_time``, ``calls``,
``start_time``, ``diff`` (the last interval) and ``average_time`` (a plain attribute like the reference's, assigned by
``toc`` = total_time / calls, 0 after ``reset``: scripts may set it, copy it or pickle the timer through ``__dict__``);
``toc(average=True)`` returns the running average, ``toc(False)`` the last interval."""
import time


class Timer(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.total_time = self.start_time = self.diff = self.average_time = 0.0
        self.calls = 0

    def tic(self):
        self.start_time = time.time()      # (wall clock, like the reference: intervals may span threads)

    def toc(self, average=True):
        self.diff = time.time() - self.start_time
        self.total_time += self.diff
        self.calls += 1
        self.average_time = self.total_time / self.calls
        return self.average_time if average else self.diff
