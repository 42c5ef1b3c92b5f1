"""Python host-side mirror of the reference's `Som` / `DataSet` / `Transformation` interface for
the training hot path, over the C ABI (capi.py -> libvsom_hip.so).  Same member names, argument
meaning and error behaviour as include/SOM.hpp:107-171, so the parity tests read like the
reference's perf harness (tests/performance/perf_tests.cpp:74-140).  The C++ mirror with the
exact signatures lives in host/ (SOM.hpp); this one exists for tests/bench plumbing.

Nothing here computes on the CPU: every search/update call goes to the HIP library.
"""
import ctypes
import enum
import math
import sys

import numpy as np

from . import capi


class WeigthDecayFunction(enum.IntEnum):      # SOM.hpp:70-75 (spelling as in the reference)
    Exponential = 0
    InverseProportional = 1
    BatchMap = 2


class Transformation:
    """Transformation.hpp:10-41; only the three built-in factories run on the device."""

    def __init__(self, kind=capi.STANDARD, names=None, name="Standard transformation"):
        self.kind, self.names, self.Name = kind, list(names or []), name

    @staticmethod
    def Standard(columnNames=()):
        return Transformation(capi.STANDARD, columnNames, "Standard transformation")

    @staticmethod
    def StandardMedianEstimator(columnNames=()):
        return Transformation(capi.MEDIAN, columnNames, "Standard median estimator transformation")

    @staticmethod
    def CombinatorialLinearRegression(columnNames=()):
        return Transformation(capi.CLR, columnNames, "Linear regression")

    def Length(self, vectorLength):            # Transformation.cpp:31-35,69-73,162-165
        return vectorLength * (vectorLength - 1) if self.kind == capi.CLR else vectorLength


class SomIndex:
    """SomIndex.hpp:7-27."""

    def __init__(self, x=0, y=0):
        self.x, self.y = int(x), int(y)

    @staticmethod
    def fromLinear(som, index):                # SomIndex.cpp:13-18 (divides by HEIGHT, Q10)
        x = index % som.getWidth()
        return SomIndex(x, (index - x) // som.getHeight())

    def getSomIndex(self, som):
        return som.getWidth() * self.y + self.x

    def getX(self):
        return self.x

    def getY(self):
        return self.y

    def __eq__(self, o):
        return self.x == o.x and self.y == o.y

    def __repr__(self):
        return f"SomIndex({self.x},{self.y})"


class ArrayDataSet:
    """DataSet over an in-memory array with IDataLoader's chunked streaming
    (IDataLoader.hpp:22-23,46; DataSet.cpp:113-160): load() yields the next <= maxLoadCount rows,
    lastBMU is zeroed on every load, the stream wraps after the last chunk."""

    def __init__(self, X, maxLoadCount=None, names=None, weights=None):
        self.X = np.ascontiguousarray(X, dtype=np.float32)
        self.depth = self.X.shape[1]
        self.maxLoadCount = maxLoadCount or self.X.shape[0]
        self._pos = 0
        self._chunks = 0
        self.data = self.X[0:0]
        self.lastBMU = np.zeros(0, np.uint64)
        self._names = list(names or [f"c{i}" for i in range(self.depth)])
        self._weights = np.ones(self.depth, np.float32) if weights is None else np.asarray(weights, np.float32)

    def vectorLength(self):
        return self.depth

    def getNames(self):
        return self._names

    def getWeights(self):
        return self._weights

    def size(self):
        return self.data.shape[0]

    def isAtStartOfDataStream(self):
        return self._pos == 0

    def hasReadWholeDataStream(self):           # DataSet.cpp:113-116
        return self._chunks > 0 and self.isAtStartOfDataStream()

    def resetStreamLoadPosition(self):          # DataSet.cpp:108-111
        self._chunks = 0

    def loadNextDataFromStream(self):           # DataSet.cpp:118-160
        if self.isAtStartOfDataStream():
            self._chunks = 0
        end = min(self._pos + self.maxLoadCount, self.X.shape[0])
        self.data = self.X[self._pos:end]
        self._pos = 0 if end >= self.X.shape[0] else end
        self.lastBMU = np.zeros(self.data.shape[0], np.uint64)   # :136-137
        self._chunks += 1

    def getData(self, i):
        return self.data[i].copy()

    def getLastBMU(self):
        return self.lastBMU


class Metrics:
    def __init__(self, n=0):
        self.MeanSquaredError = [0.0] * n
        self.DistanceError = [0.0] * n


class Som:
    """`class Som` (SOM.hpp:39-189), hot-path members only."""

    WeigthDecayFunction = WeigthDecayFunction

    def __init__(self, width, height, depth_or_dataset, transformation=None, device=0):
        self.transform = transformation or Transformation()
        if hasattr(depth_or_dataset, "vectorLength"):          # Som(w,h,DataSet,T)  SOM.hpp:78-82
            in_len = depth_or_dataset.vectorLength()
            self.transform.names = depth_or_dataset.getNames()
        else:                                                    # Som(w,h,depth,T)    SOM.hpp:83-87
            depth = int(depth_or_dataset)
            in_len = self._in_len_from_depth(depth)
        self.width, self.height = int(width), int(height)
        self.ctx = capi.Context(width, height, in_len, self.transform.kind, device=device)
        self.depth = self.ctx.depth
        self.in_len = in_len
        self.metrics = Metrics()
        self._isTraining = False
        self._verbose = False

    def _in_len_from_depth(self, depth):
        if self.transform.kind != capi.CLR:
            return depth
        J = int(round((1 + math.sqrt(1 + 4 * depth)) / 2))     # depth = J(J-1), perf_tests.cpp:338-339
        if J * (J - 1) != depth:
            raise ValueError("depth is not J*(J-1) for any J (CombinatorialLinearRegression)")
        return J

    def close(self):
        self.ctx.close()

    # ---- accessors (Som.cpp:164-212, 268-281) -------------------------------------------
    def getWidth(self):
        return self.width

    def getHeight(self):
        return self.height

    def getDepth(self):
        return self.depth

    def getIndex(self, i):
        return i.getY() * self.width + i.getX()

    def _node(self, i):
        return self.getIndex(i) if isinstance(i, SomIndex) else int(i)

    def getNeuron(self, i):
        return self.ctx.get_state(sigma=False, S=False, weight=False, hits=False)["map"][self._node(i)].copy()

    def getSigmaNeuron(self, i):
        return self.ctx.get_state(map=False, S=False, weight=False, hits=False)["sigma"][self._node(i)].copy()

    def getWeigthMap(self):
        return self.ctx.get_state(map=False, sigma=False, S=False, hits=False)["weight"]

    def getBmuHits(self):
        return self.ctx.get_state(map=False, sigma=False, S=False, weight=False)["hits"]

    def getMetrics(self):
        return self.metrics

    def isTraining(self):
        return self._isTraining

    def isCompatibleWithData(self, data):
        return self.transform.Length(data.vectorLength()) == self.depth

    def state(self):
        return self.ctx.get_state()

    def setState(self, **kw):
        self.ctx.set_state(**kw)

    def randomInitialize(self, seed, sigma):
        """Som.cpp:977-997: glibc srand/rand sequence, node-major, dim-minor."""
        libc = ctypes.CDLL(None)
        libc.srand(ctypes.c_uint(int(seed) & 0xFFFFFFFF))
        n, d = self.width * self.height, self.depth
        mod = int(np.float32(2000) * np.float32(sigma))
        r = np.fromiter((libc.rand() % mod for _ in range(n * d)), dtype=np.int64, count=n * d)
        m = ((r.astype(np.float32) - np.float32(1000.0) * np.float32(sigma)) / np.float32(1000.0)).astype(np.float32)
        self.metrics = Metrics(d)
        self.ctx.set_state(map=m.reshape(n, d), sigma=np.zeros((n, d), np.float32),
                           S=np.zeros((n, d), np.float32), weight=np.zeros(n, np.float32),
                           hits=np.zeros(n, np.uint64))

    def addBmu(self, pos):                       # Som.cpp:1189-1192
        st = self.ctx.get_state(map=False, sigma=False, S=False, weight=False)
        st["hits"][self.getIndex(pos)] += 1
        self.ctx.set_state(hits=st["hits"])

    @staticmethod
    def calculateNeighbourhoodWeight(currentX, currentY, bmuX, bmuY, currentSigma):
        return capi.neighbourhood_weight(currentX, currentY, bmuX, bmuY, currentSigma)

    # ---- search (Som.cpp:115-141, 283-309, 335-454) --------------------------------------
    def _stage_one(self, v):
        v = np.ascontiguousarray(v, dtype=np.float32).reshape(1, -1)
        self.ctx.upload_chunk(v)

    def findBmu(self, v, valid=None, weights=None):
        idx, _ = self.ctx.find_bmu(v)
        return SomIndex(idx % self.width, idx // self.width)

    def findLocalBmu(self, v, valid, lastBMUref, weights=None):
        self._stage_one(v)
        self.ctx.set_last_bmu(np.array([lastBMUref], np.uint64))
        idx, _ = self.ctx.bmu_local_batch()
        return SomIndex(int(idx[0]) % self.width, int(idx[0]) // self.width)

    def euclidianWeightedDist(self, pos, v, valid=None, weights=None):
        self._stage_one(v)
        return float(self.ctx.distances([self._node(pos)], [0])[0])

    # ---- batch training (Som.cpp:716-879) --------------------------------------------------
    def trainBatchSomEpoch(self, dataset, currentSigma, isFirst):
        self.ctx.upload_chunk(dataset.data)
        if not isFirst:
            self.ctx.set_last_bmu(dataset.lastBMU)
        mse = self.ctx.batch_epoch(currentSigma, isFirst)
        dataset.lastBMU[...] = self.ctx.get_last_bmu()
        return mse

    # ---- U-matrix (Som.cpp:143-157, 999-1111) -------------------------------------------------
    def updateUMatrix(self, weights=None):
        """Mean sigma-normalised raw distance of every node to its 3/5/8 neighbours, diagonals weighted 0.3;
        the distances on the device (vsom_distances_raw), their combination in double on the host in the
        reference's order of additions."""
        W, H = self.width, self.height
        DI = (0, 0, 1, -1, -1, 1, -1, 1)       # W, E, S(i+1), N(i-1), NW, SW, NE, SE  (:1017-1024)
        DJ = (-1, 1, 0, 0, -1, -1, 1, 1)
        nodes, nbrs, slot = [], [], {}
        for i in range(H):
            for j in range(W):
                for k in range(8):
                    ni, nj = i + DI[k], j + DJ[k]
                    if 0 <= ni < H and 0 <= nj < W:
                        slot[(i * W + j, k)] = len(nodes)
                        nodes.append(i * W + j)
                        nbrs.append(ni * W + nj)
        d = self.ctx.distances_raw(nodes, nbrs, True).astype(np.float64) if nodes else np.zeros(0)
        f = 0.3
        Wk, Ek, Sk, Nk, NWk, SWk, NEk, SEk = range(8)
        U = np.zeros(W * H, np.float64)
        for i in range(H):
            for j in range(W):
                n = i * W + j
                R = lambda k: float(d[slot[(n, k)]])   # noqa: E731
                if 0 < j < W - 1 and 0 < i < H - 1:
                    u = (R(Wk) + R(Ek) + R(Sk) + R(Nk) + R(NWk) * f + R(SWk) * f + R(NEk) * f + R(SEk) * f) / 8
                elif i == 0 and 0 < j < W - 1:
                    u = (R(Wk) + R(Ek) + R(Sk) + R(SWk) * f + R(SEk) * f) / 5
                elif i == H - 1 and 0 < j < W - 1:
                    u = (R(Wk) + R(Ek) + R(Nk) + R(NWk) * f + R(NEk) * f) / 5
                elif j == 0 and 0 < i < H - 1:
                    u = (R(Ek) + R(Sk) + R(Nk) + R(NEk) * f + R(SEk) * f) / 5
                elif j == W - 1 and 0 < i < H - 1:
                    u = (R(Wk) + R(Sk) + R(Nk) + R(NWk) * f + R(SWk) * f) / 5
                elif j == 0 and i == 0 and W > 1 and H > 1:
                    u = (R(Ek) + R(Sk) + R(SEk) * f) / 3
                elif j == W - 1 and i == 0 and W > 1 and H > 1:
                    u = (R(Wk) + R(Sk) + R(SWk) * f) / 3
                elif j == 0 and i == H - 1 and W > 1 and H > 1:
                    u = (R(Ek) + R(Nk) + R(NEk) * f) / 3
                elif j == W - 1 and i == H - 1 and W > 1 and H > 1:
                    u = (R(Wk) + R(Nk) + R(NWk) * f) / 3
                else:
                    u = 0.0
                U[n] = u
        self.uMatrix = U
        return U

    def getUMatrix(self):
        return getattr(self, "uMatrix", np.zeros(self.width * self.height, np.float64))

    def trainBatchSom(self, data, numberOfEpochs, sigma0, sigmaDecay, updateUMatrixAfterEpoch=False):
        self.metrics = Metrics(numberOfEpochs)                    # :719
        for i in range(numberOfEpochs):
            if self._verbose:
                print(f"Training VSOM epoch {i}/{numberOfEpochs}")
            sigma = sigma0 * math.exp(-sigmaDecay * float(i))     # :727
            if sigma < 1.0:                                       # :729-730
                return
            mse = np.float32(0.0)
            count = 0
            while not data.hasReadWholeDataStream():              # :735
                data.loadNextDataFromStream()
                mse = np.float32(mse + self.trainBatchSomEpoch(data, sigma, i == 0))
                count += 1
            mse = np.float32(mse / np.float32(count))             # :743
            self.metrics.MeanSquaredError[i] = mse
            data.resetStreamLoadPosition()
            if updateUMatrixAfterEpoch:
                self.updateUMatrix(data.getWeights())             # :751-752 / :1183-1184

    # ---- online training (Som.cpp:885-947, 1135-1187) ---------------------------------------
    def trainSingle(self, v, valid, weights, eta, sigma, lastBMU, weightDecayFunction):
        bmu, residual, dist, last = self.ctx.train_single(v, eta, sigma, lastBMU, int(weightDecayFunction))
        return SomIndex(bmu % self.width, bmu // self.width), residual, dist, last

    def trainBasicSom(self, data, numberOfEpochs, eta0, etaDecay, sigma0, sigmaDecay,
                      weightDecayFunction, updateUMatrixAfterEpoch=False):
        self.metrics = Metrics(numberOfEpochs)
        for i in range(numberOfEpochs):
            eta = eta0 * math.exp(-etaDecay * float(i))           # :1145
            sigma = sigma0 * math.exp(-sigmaDecay * float(i))     # :1146
            if sigma < 1.0:
                sigma = 1.0                                       # :1148-1149
            mse = np.float32(0.0)
            count = 0
            while not data.hasReadWholeDataStream():
                data.loadNextDataFromStream()
                self.ctx.upload_chunk(data.data)                  # lastBMU zeroed by the load
                # ONE running accumulator over the epoch's chunks (Som.cpp:1153,1167)
                mse = self.ctx.train_online_chunk(eta, sigma, int(weightDecayFunction), first_chunk=(count == 0))
                data.lastBMU[...] = self.ctx.get_last_bmu()
                count += 1
            mse = np.float32(mse / np.float32(count))             # :1175
            self.metrics.MeanSquaredError[i] = mse
            data.resetStreamLoadPosition()
            if updateUMatrixAfterEpoch:
                self.updateUMatrix(data.getWeights())             # :751-752 / :1183-1184

    def train(self, data, numberOfEpochs, eta0, etaDecay, sigma0, sigmaDecay, weightDecayFunction,
              updateUMatrixAfterEpoch=False):
        """Som.cpp:1113-1132: dispatch; exceptions are printed to stderr, not raised."""
        self._isTraining = True
        try:
            if int(weightDecayFunction) == WeigthDecayFunction.BatchMap:
                self.trainBatchSom(data, numberOfEpochs, sigma0, sigmaDecay, updateUMatrixAfterEpoch)
            else:
                self.trainBasicSom(data, numberOfEpochs, eta0, etaDecay, sigma0, sigmaDecay,
                                   weightDecayFunction, updateUMatrixAfterEpoch)
        except Exception as e:      # noqa: BLE001  (mirrors the reference's catch-and-print)
            print(e, file=sys.stderr)
        self._isTraining = False
