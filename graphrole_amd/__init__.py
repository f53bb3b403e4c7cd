"""
graphrole_amd -- MI355X-native ReFeX / RolX engine, drop-in for GraphRole's
``RecursiveFeatureExtractor`` and ``RoleExtractor`` (reference: graphrole/__init__.py:1-2).
"""
__version__ = '0.1.0'


def __getattr__(name):
    # lazy: importing the package must not require a GPU (CPU-side tests import sub-modules)
    if name == 'RecursiveFeatureExtractor':
        from .features.extract import RecursiveFeatureExtractor
        return RecursiveFeatureExtractor
    if name == 'RoleExtractor':
        from .roles.extract import RoleExtractor
        return RoleExtractor
    raise AttributeError(name)
