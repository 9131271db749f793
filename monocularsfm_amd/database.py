"""Python twin of the reference's SQLite surface on the ComputeMatches path.

Mirrors /root/reference/src/Database/Database.cpp: same tables (:710-764), pragmas (:299-302),
user_version (:926-928), blob codec (rows, cols, raw little-endian data; :82-88, :230-278),
pair-id encoding and swap canonicalisation (:631-694).  Method names follow Database.h.
Used by the tests, the bench fixtures and the multi-GPU driver; the C++ host
(monocularsfm_amd/host/Database.cpp) implements the same surface for the ComputeMatches CLI.
"""
import sqlite3

import numpy as np

kMaxNumImages = 10000
kSchemaVersion = 1


def SwapImagePair(image_id1, image_id2):
    return image_id1 > image_id2


def ImagePairToPairId(image_id1, image_id2):
    assert 0 <= image_id1 < kMaxNumImages and 0 <= image_id2 < kMaxNumImages
    if SwapImagePair(image_id1, image_id2):
        return kMaxNumImages * image_id2 + image_id1
    return kMaxNumImages * image_id1 + image_id2


def PairIdToImagePair(pair_id):
    image_id2 = pair_id % kMaxNumImages
    image_id1 = (pair_id - image_id2) // kMaxNumImages
    return image_id1, image_id2


class Database:
    def __init__(self, path=None):
        self.db = None
        if path is not None:
            self.Open(path)

    # -- open / close / transactions ------------------------------------------------------
    def Open(self, path):
        self.db = sqlite3.connect(path, isolation_level=None)  # autocommit; explicit BEGIN/END
        c = self.db
        c.execute("PRAGMA synchronous=OFF")
        c.execute("PRAGMA journal_mode=WAL")
        c.execute("PRAGMA temp_store=MEMORY")
        c.execute("PRAGMA foreign_keys=ON")
        c.execute("CREATE TABLE IF NOT EXISTS images"
                  "(  image_id  INTEGER PRIMARY KEY AUTOINCREMENT   NOT NULL,"
                  "   name      TEXT                                NOT NULL UNIQUE)")
        for table in ("keypoints", "colors", "descriptors"):
            c.execute("CREATE TABLE IF NOT EXISTS %s"
                      "  (image_id    INTEGER    PRIMARY KEY    NOT NULL,"
                      "   rows        INTEGER                   NOT NULL,"
                      "   cols        INTEGER                   NOT NULL,"
                      "   data        BLOB,"
                      "FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE)" % table)
        c.execute("CREATE TABLE IF NOT EXISTS matches"
                  "   (pair_id    INTEGER    PRIMARY KEY    NOT NULL,"
                  "    rows       INTEGER                   NOT NULL,"
                  "    cols       INTEGER                   NOT NULL,"
                  "    data       BLOB);")
        c.execute("PRAGMA user_version = %d;" % kSchemaVersion)

    def Close(self):
        if self.db is not None:
            self.db.close()
            self.db = None

    def BeginTransaction(self):
        self.db.execute("BEGIN TRANSACTION")

    def EndTransaction(self):
        self.db.execute("END TRANSACTION")

    # -- existence / counts ---------------------------------------------------------------
    def _exists(self, sql, key):
        return self.db.execute(sql, (key,)).fetchone() is not None

    def ExistImageById(self, image_id):
        return self._exists("SELECT 1 FROM images WHERE image_id = ?;", image_id)

    def ExistKeyPoints(self, image_id):
        return self._exists("SELECT 1 FROM keypoints WHERE image_id = ?;", image_id)

    def ExistDescriptors(self, image_id):
        return self._exists("SELECT 1 FROM descriptors WHERE image_id = ?;", image_id)

    def ExistMatches(self, image_id1, image_id2):
        return self._exists("SELECT 1 FROM matches WHERE pair_id = ?;", ImagePairToPairId(image_id1, image_id2))

    def NumImages(self):
        return self.db.execute("SELECT COUNT(*) FROM images;").fetchone()[0]

    def NumMatches(self, image_id1, image_id2):
        r = self.db.execute("SELECT rows FROM matches WHERE pair_id = ?;",
                            (ImagePairToPairId(image_id1, image_id2),)).fetchone()
        return r[0]

    # -- blob codec -------------------------------------------------------------------------
    @staticmethod
    def _read_blob(row, dtype):
        rows, cols, data = row
        a = np.frombuffer(data if data is not None else b"", dtype=dtype)
        assert a.size == rows * cols, "blob size does not match rows*cols"
        return a.reshape(rows, cols).copy()

    # -- reads ------------------------------------------------------------------------------
    def ReadAllImages(self):
        return [(int(i), n) for i, n in self.db.execute("SELECT * FROM images;")]

    def ReadKeyPoints(self, image_id):
        """n x 4 float32 (x, y, size, angle)."""
        row = self.db.execute("SELECT rows, cols, data FROM keypoints WHERE image_id = ?;", (image_id,)).fetchone()
        return self._read_blob(row, np.float32)

    def ReadDescriptors(self, image_id):
        row = self.db.execute("SELECT rows, cols, data FROM descriptors WHERE image_id = ?;", (image_id,)).fetchone()
        return self._read_blob(row, np.float32)

    def ReadMatches(self, image_id1, image_id2):
        """m x 2 int32 (queryIdx in image_id1, trainIdx in image_id2)."""
        row = self.db.execute("SELECT rows, cols, data FROM matches WHERE pair_id = ?;",
                              (ImagePairToPairId(image_id1, image_id2),)).fetchone()
        m = self._read_blob(row, np.int32)
        if SwapImagePair(image_id1, image_id2):
            m = m[:, ::-1].copy()
        return m

    def ReadAllMatches(self):
        out = []
        for pair_id, rows, cols, data in self.db.execute("SELECT * FROM matches WHERE rows > 0;"):
            out.append((int(pair_id), self._read_blob((rows, cols, data), np.int32)))
        return out

    # -- writes -----------------------------------------------------------------------------
    def WriteImage(self, image_id, name, use_image_id=True):
        if use_image_id:
            self.db.execute("INSERT INTO images(image_id, name) VALUES(?, ?);", (int(image_id), name))
            return int(image_id)
        cur = self.db.execute("INSERT INTO images(image_id, name) VALUES(?, ?);", (None, name))
        return cur.lastrowid

    def _write_blob(self, table, key, arr, dtype):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        rows, cols = arr.shape
        self.db.execute("INSERT INTO %s VALUES(?, ?, ?, ?);" % table, (int(key), rows, cols, arr.tobytes()))

    def WriteKeyPoints(self, image_id, keypoints):
        self._write_blob("keypoints(image_id, rows, cols, data)", image_id, np.asarray(keypoints).reshape(-1, 4), np.float32)

    def WriteDescriptors(self, image_id, descriptors):
        self._write_blob("descriptors(image_id, rows, cols, data)", image_id, descriptors, np.float32)

    def WriteMatches(self, image_id1, image_id2, matches):
        """matches: m x 2 (queryIdx, trainIdx); column 0 is stored for the smaller image id."""
        m = np.asarray(matches, dtype=np.int32).reshape(-1, 2)
        if SwapImagePair(image_id1, image_id2):
            m = m[:, ::-1]
        self._write_blob("matches(pair_id, rows, cols, data)", ImagePairToPairId(image_id1, image_id2), m, np.int32)


def write_synthetic_database(path, descriptors, keypoints=None, names=None):
    """Create a database the way FeatureExtraction leaves it: dense ids 0..N-1
    (src/Feature/FeatureExtraction.cpp:74-76), keypoints n x 4 and descriptors n x 128 float32."""
    from . import synth
    db = Database(path)
    db.BeginTransaction()
    for i, d in enumerate(descriptors):
        db.WriteImage(i, names[i] if names else "image_%05d.jpg" % i)
        k = keypoints[i] if keypoints is not None else synth.keypoints(len(d), seed=1000 + i)
        db.WriteKeyPoints(i, k)
        db.WriteDescriptors(i, np.asarray(d, dtype=np.float32))
    db.EndTransaction()
    db.Close()
