"""tic/toc timer with the attributes T-CNN scripts read (reference utils/timer.py:10-32):
total_time, calls, start_time, diff, average_time."""
import time


class Timer(object):
    def __init__(self):
        self.total_time = 0.
        self.calls = 0
        self.start_time = 0.
        self.diff = 0.
        self.average_time = 0.

    def tic(self):
        self.start_time = time.time()

    def toc(self, average=True):
        now = time.time()
        self.diff = now - self.start_time
        self.calls += 1
        self.total_time += self.diff
        self.average_time = self.total_time / self.calls
        return self.average_time if average else self.diff
