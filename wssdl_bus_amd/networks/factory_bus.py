"""get_network (reference: code/lib/networks/factory_bus.py:28-44)."""
from .Resnet_train_bus import Resnet_train_bus
from .VGGnet_train_bus import VGGnet_train_bus


def get_network(name, net_depth=50, dataset='SNUBH', norm_type='BN'):
    if name == 'VGGnet_train':
        return VGGnet_train_bus(dataset)
    elif name == 'VGGnet_train_alter':
        return VGGnet_train_bus(dataset, alter=True)
    elif name == 'Resnet_train':
        return Resnet_train_bus(net_depth, dataset, norm_type)
    elif name == 'Resnet_train_alter':
        return Resnet_train_bus(net_depth, dataset, norm_type, alter=True)
    raise KeyError('Unknown network: {}'.format(name))
