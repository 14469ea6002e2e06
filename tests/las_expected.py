"""Expected values of the reference's LAS fixtures — data restated from pasture-io/src/las/test_util.rs:46-188."""
import numpy as np

N = 10
POSITIONS = np.array([[i, i, i] for i in range(N)], dtype=np.float64)                       # :50-63
BOUNDS = ((0.0, 0.0, 0.0), (9.0, 9.0, 9.0))                                                  # :46-48
INTENSITIES = np.array([255 * i for i in range(N)], dtype=np.uint16)                        # :65-78
RETURN_NUMBERS = np.array([0, 1, 2, 3, 4, 5, 6, 7, 0, 1], dtype=np.uint8)                   # :80-82
RETURN_NUMBERS_EXTENDED = np.arange(N, dtype=np.uint8)                                      # :84-86
NUMBER_OF_RETURNS = np.array([0, 1, 2, 3, 4, 5, 6, 7, 0, 1], dtype=np.uint8)                # :88-90
NUMBER_OF_RETURNS_EXTENDED = np.arange(N, dtype=np.uint8)                                   # :92-94
CLASSIFICATION_FLAGS = np.arange(N, dtype=np.uint8)                                         # :96-98
SCANNER_CHANNELS = np.array([0, 1, 2, 3, 0, 1, 2, 3, 0, 1], dtype=np.uint8)                 # :100-102
SCAN_DIRECTION_FLAGS = np.array([0, 1] * 5, dtype=np.uint8)                                 # :104-106
EDGE_OF_FLIGHT_LINES = np.array([0, 1] * 5, dtype=np.uint8)                                 # :108-110
CLASSIFICATIONS = np.arange(N, dtype=np.uint8)                                              # :112-114
SCAN_ANGLE_RANKS = np.arange(N, dtype=np.int8)                                              # :116-118
SCAN_ANGLES_EXTENDED = np.arange(N, dtype=np.int16)                                         # :120-122
USER_DATA = np.arange(N, dtype=np.uint8)                                                    # :124-126
POINT_SOURCE_IDS = np.arange(N, dtype=np.uint16)                                            # :128-130
GPS_TIMES = np.arange(1, N + 1, dtype=np.float64)                                           # :132-134
COLORS = np.array([[i, (i + 1) << 4, (i + 2) << 8] for i in range(N)], dtype=np.uint16)     # :136-149
NIRS = np.arange(N, dtype=np.uint16)                                                        # :151-153
WAVEPACKET_INDEX = np.arange(N, dtype=np.uint8)                                             # :155-157
WAVEPACKET_OFFSET = np.arange(N, dtype=np.uint64)                                           # :159-161
WAVEPACKET_SIZE = np.arange(N, dtype=np.uint32)                                             # :163-165
WAVEPACKET_LOCATION = np.arange(N, dtype=np.float32)                                        # :167-169
WAVEPACKET_PARAMETERS = np.array([[i + 1, i + 2, i + 3] for i in range(N)], dtype=np.float32)  # :171-184
