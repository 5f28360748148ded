"""argparse-namespace stand-ins for the loss flags of train.py:21-66 (read by utils.get_loss, utils.py:9-20)."""


class LossArgs:
    def __init__(self, kldiv=True, cc=False, sim=False, l1=False, kldiv_coeff=1.0, cc_coeff=-1.0, sim_coeff=-1.0, l1_coeff=1.0,
                 batch_size=2, **extra):
        self.kldiv, self.cc, self.sim, self.l1 = kldiv, cc, sim, l1
        self.kldiv_coeff, self.cc_coeff, self.sim_coeff, self.l1_coeff = kldiv_coeff, cc_coeff, sim_coeff, l1_coeff
        self.batch_size = batch_size
        self.__dict__.update(extra)
