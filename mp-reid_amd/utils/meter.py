"""``AverageMeter`` (same attributes as the reference's utils/meter.py: val, avg, sum, count; update(val, n)).
Only the training loop of the reference uses it; it is here because ``processor/processor.py`` imports it."""


class AverageMeter:
    __slots__ = ("val", "sum", "count")

    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.sum, self.count = 0, 0, 0

    @property
    def avg(self):
        """weighted mean of everything passed to update() since the last reset (0 before the first update)"""
        return self.sum / self.count if self.count else 0

    def update(self, val, n=1):
        self.val = val
        self.count += n
        self.sum += n * val
