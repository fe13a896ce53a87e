"""ControlUnit: the abstract controller interface of drone/onboard_computer.py:34-42."""
import abc


class ControlUnit(abc.ABC):
    def __init__(self):
        pass

    @abc.abstractmethod
    def translate_input_to_thrust(self, *args, **kwargs):
        raise NotImplementedError
