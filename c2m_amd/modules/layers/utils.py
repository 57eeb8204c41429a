"""init_weights and parameter filters (reference: src/modules/layers/utils.py:4-33)."""
from torch.nn import init


def weight_parameters(module):
    return [p for n, p in module.named_parameters() if 'weight' in n]


def bias_parameters(module):
    return [p for n, p in module.named_parameters() if 'bias' in n]


def init_weights(net, init_type='normal', init_gain=0.02):
    fillers = {
        'normal': lambda w: init.normal_(w, 0.0, init_gain),
        'xavier': lambda w: init.xavier_normal_(w, gain=init_gain),
        'kaiming': lambda w: init.kaiming_normal_(w, a=0, mode='fan_in'),
        'orthogonal': lambda w: init.orthogonal_(w, gain=init_gain),
    }
    if init_type not in fillers:
        raise NotImplementedError('initialization method [%s] is not implemented' % init_type)

    def visit(m):
        name = m.__class__.__name__
        if hasattr(m, 'weight') and ('Conv' in name or 'Linear' in name):
            fillers[init_type](m.weight.data)
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias.data, 0.0)
        elif 'BatchNorm2d' in name:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)

    net.apply(visit)
