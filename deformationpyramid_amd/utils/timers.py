"""Wall-clock timers with the reference's protocol (/root/reference/utils/tiktok.py:10-75):
Timers().tic(key) / .toc(key) / .tictoc(key, diff) / .get_strings() / .get_avg(key).
register(timer=...) accepts one of these and hands it back (registration.py:262)."""
import time
from collections import defaultdict


class Timer:
    def __init__(self):
        self.reset()

    def reset(self):
        self.total_time, self.calls, self.start_time, self.diff = 0.0, 0, 0.0, 0.0

    def tic(self):
        self.start_time = time.time()

    def toc(self, average=True):
        self.tictoc(time.time() - self.start_time)

    def tictoc(self, diff):
        self.diff = diff
        self.total_time += diff
        self.calls += 1

    def total(self):
        return self.total_time

    def avg(self):
        return self.total_time / float(self.calls)


class Timers:
    def __init__(self):
        self.timers = defaultdict(Timer)

    def tic(self, key):
        self.timers[key].tic()

    def toc(self, key):
        self.timers[key].toc()

    def tictoc(self, key, diff):
        self.timers[key].tictoc(diff)

    def get_avg(self, key):
        return self.timers[key].avg()

    def get_strings(self):
        return ["{:}: \t  average {:.4f},  total {:.4f} ,\t calls {:}".format(k.ljust(30), v.avg(), v.total_time, v.calls)
                for k, v in self.timers.items()]

    def print(self, key=None):
        if key is None:
            for line in self.get_strings():
                print(line)
        else:
            print("Average time for {:}: {:}".format(key, self.timers[key].avg()))
