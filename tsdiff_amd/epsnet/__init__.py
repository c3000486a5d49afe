"""Mirror of the reference registry `models/epsnet/__init__.py:1-15`."""


def get_model(config):
    network = config["network"] if isinstance(config, dict) else config.network
    if network == "condensenc":
        from .condensenc import CondenseEncoderEpsNetwork
        return CondenseEncoderEpsNetwork(config)
    if network == "dualenc":  # GeoDiff legacy network (configs/geodiff_legacy/*.yml)
        from .dualenc import DualEncoderEpsNetwork
        return DualEncoderEpsNetwork(config)
    # `dualenc_general` imports a file that does not exist in the reference.
    raise NotImplementedError("Unknown network: %s" % network)
