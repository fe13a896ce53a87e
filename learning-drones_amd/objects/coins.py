"""Coin: mirror of objects/coins.py:4-25.  Host-side description of a target; the batched kernels
consume coins as rows (x, y, z, radius, value) of the device coin table (params.coin_table)."""
import numpy as np


class Coin:
    def __init__(self, position, radius, value=100):
        position = np.asarray(position)
        if position.shape != (3, 1):
            raise AssertionError(position.shape)             # the reference asserts the (3, 1) shape
        self.position = position
        self.x, self.y, self.z = (position[i, 0] for i in range(3))
        self.radius = radius
        self.value = value

    def distance_drone_to_coin(self, drone_position):
        return np.linalg.norm(self.position - drone_position)

    def is_in_vicinity(self, drone_position):
        if np.asarray(drone_position).shape != (3, 1):
            raise AssertionError(np.asarray(drone_position).shape)
        return self.distance_drone_to_coin(drone_position) < self.radius      # strict, coins.py:20

    def row(self):
        return [float(self.x), float(self.y), float(self.z), float(self.radius), float(self.value)]
