"""Running-mean training log (the role of VO_Module/droid_slam/logger.py): metrics are summed per step and their means
printed every SUM_FREQ steps with the step count and the scheduler's learning rate.  TensorBoard is written only if it is
installed (it is not in this image); the printed line has the reference's layout."""
SUM_FREQ = 100


class Logger:
    def __init__(self, name, scheduler, sum_freq=SUM_FREQ, out=print):
        self.name, self.scheduler, self.sum_freq, self.out = name, scheduler, sum_freq, out
        self.total_steps = 0
        self.running = {}
        self.writer = None
        self.history = []            # (step, {metric: mean}) of every printed line

    def _flush(self):
        if self.writer is None:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter("runs/%s" % self.name)
            except Exception:
                self.writer = False
            self.out(list(self.running))
        lr = self.scheduler.get_last_lr()[-1] if self.scheduler is not None else 0.0
        means = {k: v / self.sum_freq for k, v in self.running.items()}
        self.out("[{:6d}, {:10.7f}] ".format(self.total_steps + 1, lr) + ("{:10.4f}, " * len(means)).format(*means.values()))
        if self.writer:
            for k, v in means.items():
                self.writer.add_scalar(k, v, self.total_steps)
        self.history.append((self.total_steps + 1, means))
        self.running = {}

    def push(self, metrics):
        for k, v in metrics.items():
            self.running[k] = self.running.get(k, 0.0) + float(v)
        if self.total_steps % self.sum_freq == self.sum_freq - 1:
            self._flush()
        self.total_steps += 1

    def write_dict(self, results):
        if self.writer:
            for k, v in results.items():
                self.writer.add_scalar(k, v, self.total_steps)

    def close(self):
        if self.writer:
            self.writer.close()
