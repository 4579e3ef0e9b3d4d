// Minimal restatement of the open_vins ov_type classes the update path touches (ext ov_core/src/types/, not vendored
// in the reference tree; semantics per SURVEY.md Appendix A): Type, Vec, JPLQuat, PoseJPL, IMU, Landmark.
// Dense storage is ov_plane::DenseMatrix (same data()/rows()/cols()/operator() surface as Eigen::MatrixXd, which is
// not installed here): switching the typedefs below to Eigen's types is the only change a build with Eigen needs.
#pragma once
#include <cassert>
#include <cmath>
#include <cstddef>
#include <memory>
#include <unordered_map>
#include <vector>

namespace ov_plane {

class DenseMatrix {  // column-major, like Eigen::MatrixXd
public:
  DenseMatrix() : r_(0), c_(0) {}
  DenseMatrix(int r, int c) : r_(r), c_(c), d_((size_t)r * c, 0.0) {}
  static DenseMatrix Zero(int r, int c) { return DenseMatrix(r, c); }
  static DenseMatrix Identity(int r, int c) {
    DenseMatrix m(r, c);
    for (int i = 0; i < (r < c ? r : c); ++i) m(i, i) = 1.0;
    return m;
  }
  int rows() const { return r_; }
  int cols() const { return c_; }
  double *data() { return d_.data(); }
  const double *data() const { return d_.data(); }
  double &operator()(int i, int j) { return d_[(size_t)j * r_ + i]; }
  double operator()(int i, int j) const { return d_[(size_t)j * r_ + i]; }
  double &operator()(int i) { return d_[i]; }
  double operator()(int i) const { return d_[i]; }
  void resize(int r, int c) {
    r_ = r;
    c_ = c;
    d_.assign((size_t)r * c, 0.0);
  }
  DenseMatrix block(int i0, int j0, int nr, int nc) const {
    DenseMatrix b(nr, nc);
    for (int j = 0; j < nc; ++j)
      for (int i = 0; i < nr; ++i) b(i, j) = (*this)(i0 + i, j0 + j);
    return b;
  }

private:
  int r_, c_;
  std::vector<double> d_;
};
typedef DenseMatrix MatrixXd;
typedef DenseMatrix VectorXd;  // n x 1

}  // namespace ov_plane

namespace ov_type {
using ov_plane::MatrixXd;
using ov_plane::VectorXd;

// ---- ext quat_ops.h (JPL) ----
inline void quat_2_Rot(const double q[4], double R[9]) {  // row-major
  const double x = q[0], y = q[1], z = q[2], w = q[3], a = 2.0 * w * w - 1.0;
  R[0] = a + 2 * x * x;
  R[1] = 2 * w * z + 2 * x * y;
  R[2] = -2 * w * y + 2 * x * z;
  R[3] = -2 * w * z + 2 * y * x;
  R[4] = a + 2 * y * y;
  R[5] = 2 * w * x + 2 * y * z;
  R[6] = 2 * w * y + 2 * z * x;
  R[7] = -2 * w * x + 2 * z * y;
  R[8] = a + 2 * z * z;
}
inline void quat_multiply(const double q[4], const double p[4], double o[4]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  double r[4] = {w * p[0] + z * p[1] - y * p[2] + x * p[3], -z * p[0] + w * p[1] + x * p[2] + y * p[3],
                 y * p[0] - x * p[1] + w * p[2] + z * p[3], -x * p[0] - y * p[1] - z * p[2] + w * p[3]};
  if (r[3] < 0)
    for (double &v : r) v = -v;
  const double n = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  for (int k = 0; k < 4; ++k) o[k] = r[k] / n;
}

class Type {
public:
  explicit Type(int size) : _size(size) {}
  virtual ~Type() {}
  virtual void set_local_id(int new_id) { _id = new_id; }
  int id() const { return _id; }
  int size() const { return _size; }
  virtual void update(const VectorXd &dx) = 0;
  virtual const VectorXd &value() const { return _value; }
  virtual const VectorXd &fej() const { return _fej; }
  virtual void set_value(const VectorXd &v) { _value = v; }
  virtual void set_fej(const VectorXd &v) { _fej = v; }
  virtual std::shared_ptr<Type> clone() = 0;
  virtual std::shared_ptr<Type> check_if_subvariable(const std::shared_ptr<Type> check) { return nullptr; }

protected:
  VectorXd _fej, _value;
  int _id = -1;
  int _size = -1;
};

class Vec : public Type {
public:
  explicit Vec(int dim) : Type(dim) {
    _value = VectorXd::Zero(dim, 1);
    _fej = VectorXd::Zero(dim, 1);
  }
  void update(const VectorXd &dx) override {
    assert(dx.rows() == _size);
    for (int i = 0; i < _size; ++i) _value(i) += dx(i);
  }
  std::shared_ptr<Type> clone() override {
    auto c = std::make_shared<Vec>(_size);
    c->set_value(value());
    c->set_fej(fej());
    return c;
  }
};

class JPLQuat : public Type {
public:
  JPLQuat() : Type(3) {
    VectorXd q0 = VectorXd::Zero(4, 1);
    q0(3) = 1.0;
    set_value(q0);
    set_fej(q0);
  }
  void update(const VectorXd &dx) override {  // dq = quatnorm([dth/2, 1]); q <- dq (x) q
    double dq[4] = {0.5 * dx(0), 0.5 * dx(1), 0.5 * dx(2), 1.0};
    const double n = std::sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]);
    for (double &v : dq) v /= n;
    double o[4];
    quat_multiply(dq, _value.data(), o);
    VectorXd nv(4, 1);
    for (int k = 0; k < 4; ++k) nv(k) = o[k];
    set_value(nv);
  }
  void set_value(const VectorXd &v) override {
    _value = v;
    quat_2_Rot(_value.data(), _R);
  }
  void set_fej(const VectorXd &v) override {
    _fej = v;
    quat_2_Rot(_fej.data(), _Rfej);
  }
  const double *Rot() const { return _R; }        // row-major 3x3
  const double *Rot_fej() const { return _Rfej; }
  std::shared_ptr<Type> clone() override {
    auto c = std::make_shared<JPLQuat>();
    c->set_value(value());
    c->set_fej(fej());
    return c;
  }

protected:
  double _R[9], _Rfej[9];
};

class PoseJPL : public Type {
public:
  PoseJPL() : Type(6) {
    _q = std::make_shared<JPLQuat>();
    _p = std::make_shared<Vec>(3);
    _value = VectorXd::Zero(7, 1);
    _value(3) = 1.0;
    _fej = _value;
  }
  void set_local_id(int new_id) override {
    _id = new_id;
    _q->set_local_id(new_id);
    _p->set_local_id(new_id + (new_id != -1 ? 3 : 0));
  }
  void update(const VectorXd &dx) override {
    VectorXd a(3, 1), b(3, 1);
    for (int k = 0; k < 3; ++k) {
      a(k) = dx(k);
      b(k) = dx(3 + k);
    }
    _q->update(a);
    _p->update(b);
    sync_value();
  }
  void set_value(const VectorXd &v) override {
    VectorXd q(4, 1), p(3, 1);
    for (int k = 0; k < 4; ++k) q(k) = v(k);
    for (int k = 0; k < 3; ++k) p(k) = v(4 + k);
    _q->set_value(q);
    _p->set_value(p);
    _value = v;
  }
  void set_fej(const VectorXd &v) override {
    VectorXd q(4, 1), p(3, 1);
    for (int k = 0; k < 4; ++k) q(k) = v(k);
    for (int k = 0; k < 3; ++k) p(k) = v(4 + k);
    _q->set_fej(q);
    _p->set_fej(p);
    _fej = v;
  }
  const double *Rot() const { return _q->Rot(); }
  const double *Rot_fej() const { return _q->Rot_fej(); }
  const double *pos() const { return _p->value().data(); }
  const double *pos_fej() const { return _p->fej().data(); }
  const double *quat() const { return _q->value().data(); }
  const double *quat_fej() const { return _q->fej().data(); }
  std::shared_ptr<JPLQuat> q() { return _q; }
  std::shared_ptr<Vec> p() { return _p; }
  std::shared_ptr<Type> clone() override {
    auto c = std::make_shared<PoseJPL>();
    c->set_value(value());
    c->set_fej(fej());
    return c;
  }
  std::shared_ptr<Type> check_if_subvariable(const std::shared_ptr<Type> check) override {
    if (check == _q) return _q;
    if (check == _p) return _p;
    return nullptr;
  }

protected:
  void sync_value() {
    for (int k = 0; k < 4; ++k) _value(k) = _q->value()(k);
    for (int k = 0; k < 3; ++k) _value(4 + k) = _p->value()(k);
  }
  std::shared_ptr<JPLQuat> _q;
  std::shared_ptr<Vec> _p;
};

class IMU : public Type {  // [q p v bg ba], error state 15
public:
  IMU() : Type(15) {
    _pose = std::make_shared<PoseJPL>();
    _v = std::make_shared<Vec>(3);
    _bg = std::make_shared<Vec>(3);
    _ba = std::make_shared<Vec>(3);
    _value = VectorXd::Zero(16, 1);
    _value(3) = 1.0;
    _fej = _value;
  }
  void set_local_id(int new_id) override {
    _id = new_id;
    _pose->set_local_id(new_id);
    _v->set_local_id(_pose->id() + (new_id != -1 ? _pose->size() : 0));
    _bg->set_local_id(_v->id() + (new_id != -1 ? _v->size() : 0));
    _ba->set_local_id(_bg->id() + (new_id != -1 ? _bg->size() : 0));
  }
  void update(const VectorXd &dx) override {
    VectorXd a(6, 1), b(3, 1), c(3, 1), d(3, 1);
    for (int k = 0; k < 6; ++k) a(k) = dx(k);
    for (int k = 0; k < 3; ++k) {
      b(k) = dx(6 + k);
      c(k) = dx(9 + k);
      d(k) = dx(12 + k);
    }
    _pose->update(a);
    _v->update(b);
    _bg->update(c);
    _ba->update(d);
  }
  std::shared_ptr<PoseJPL> pose() { return _pose; }
  std::shared_ptr<JPLQuat> q() { return _pose->q(); }
  std::shared_ptr<Vec> p() { return _pose->p(); }
  std::shared_ptr<Vec> v() { return _v; }
  std::shared_ptr<Vec> bg() { return _bg; }
  std::shared_ptr<Vec> ba() { return _ba; }
  const double *vel() const { return _v->value().data(); }
  const double *Rot() const { return _pose->Rot(); }
  const double *Rot_fej() const { return _pose->Rot_fej(); }
  const double *quat() const { return _pose->quat(); }
  const double *quat_fej() const { return _pose->quat_fej(); }
  const double *pos() const { return _pose->pos(); }
  const double *pos_fej() const { return _pose->pos_fej(); }
  const double *vel_fej() const { return _v->fej().data(); }
  const double *bias_g() const { return _bg->value().data(); }
  const double *bias_a() const { return _ba->value().data(); }
  // 16-vector [q p v bg ba] like ext ov_type::IMU::set_value / set_fej
  void set_value(const VectorXd &x) override {
    split(x, true);
    _value = x;
  }
  void set_fej(const VectorXd &x) override {
    split(x, false);
    _fej = x;
  }
  const VectorXd &value() const override {
    gather(true);
    return _value;
  }
  const VectorXd &fej() const override {
    gather(false);
    return _fej;
  }
  std::shared_ptr<Type> clone() override {
    auto c = std::make_shared<IMU>();
    c->pose()->set_value(_pose->value());
    c->pose()->set_fej(_pose->fej());
    return c;
  }
  std::shared_ptr<Type> check_if_subvariable(const std::shared_ptr<Type> check) override {
    if (check == _pose) return _pose;
    if (check == _pose->check_if_subvariable(check)) return _pose->check_if_subvariable(check);
    if (check == _v) return _v;
    if (check == _bg) return _bg;
    if (check == _ba) return _ba;
    return nullptr;
  }

protected:
  void split(const VectorXd &x, bool val) {
    VectorXd a(7, 1), b(3, 1), c(3, 1), d(3, 1);
    for (int k = 0; k < 7; ++k) a(k) = x(k);
    for (int k = 0; k < 3; ++k) {
      b(k) = x(7 + k);
      c(k) = x(10 + k);
      d(k) = x(13 + k);
    }
    if (val) {
      _pose->set_value(a);
      _v->set_value(b);
      _bg->set_value(c);
      _ba->set_value(d);
    } else {
      _pose->set_fej(a);
      _v->set_fej(b);
      _bg->set_fej(c);
      _ba->set_fej(d);
    }
  }
  void gather(bool val) const {  // sub-variables are updated individually (IMU::update), the composite is rebuilt on read
    VectorXd &o = val ? const_cast<VectorXd &>(_value) : const_cast<VectorXd &>(_fej);
    const VectorXd &a = val ? _pose->value() : _pose->fej(), &b = val ? _v->value() : _v->fej();
    const VectorXd &c = val ? _bg->value() : _bg->fej(), &d = val ? _ba->value() : _ba->fej();
    for (int k = 0; k < 7; ++k) o(k) = a(k);
    for (int k = 0; k < 3; ++k) {
      o(7 + k) = b(k);
      o(10 + k) = c(k);
      o(13 + k) = d(k);
    }
  }
  std::shared_ptr<PoseJPL> _pose;
  std::shared_ptr<Vec> _v, _bg, _ba;
};

// ext ov_type::LandmarkRepresentation (types/LandmarkRepresentation.h)
struct LandmarkRepresentation {
  enum Representation {
    GLOBAL_3D = 0,
    GLOBAL_FULL_INVERSE_DEPTH = 1,
    ANCHORED_3D = 2,
    ANCHORED_FULL_INVERSE_DEPTH = 3,
    ANCHORED_MSCKF_INVERSE_DEPTH = 4,
    ANCHORED_INVERSE_DEPTH_SINGLE = 5,
    UNKNOWN = 6
  };
  static bool is_relative_representation(Representation r) {
    return r == ANCHORED_3D || r == ANCHORED_FULL_INVERSE_DEPTH || r == ANCHORED_MSCKF_INVERSE_DEPTH || r == ANCHORED_INVERSE_DEPTH_SINGLE;
  }
};

// ext ov_type::Landmark (types/Landmark.h/.cpp): a 3-dof (1-dof for ANCHORED_INVERSE_DEPTH_SINGLE) Vec holding the
// landmark in one of the six representations; get_xyz / set_from_xyz convert to / from the position in the global
// (GLOBAL_*) or anchor camera (ANCHORED_*) frame
class Landmark : public Vec {
public:
  explicit Landmark(int dim) : Vec(dim) {}
  size_t _featid = 0;
  int _unique_camera_id = -1;
  int _anchor_cam_id = -1;
  double _anchor_clone_timestamp = -1;
  bool has_had_anchor_change = false;
  bool should_marg = false;
  double uv_norm_zero[3] = {0, 0, 1}, uv_norm_zero_fej[3] = {0, 0, 1};  // bearing of the single-depth representation
  LandmarkRepresentation::Representation _feat_representation = LandmarkRepresentation::GLOBAL_3D;

  void get_xyz(bool getfej, double out[3]) const {
    const VectorXd &v = getfej ? fej() : value();
    typedef LandmarkRepresentation LR;
    switch (_feat_representation) {
    case LR::GLOBAL_3D:
    case LR::ANCHORED_3D:
      for (int k = 0; k < 3; ++k) out[k] = v(k);
      return;
    case LR::GLOBAL_FULL_INVERSE_DEPTH:
    case LR::ANCHORED_FULL_INVERSE_DEPTH: {
      const double th = v(0), phi = v(1), rho = v(2);
      out[0] = (1 / rho) * std::cos(th) * std::sin(phi);
      out[1] = (1 / rho) * std::sin(th) * std::sin(phi);
      out[2] = (1 / rho) * std::cos(phi);
      return;
    }
    case LR::ANCHORED_MSCKF_INVERSE_DEPTH:
      out[0] = (1 / v(2)) * v(0);
      out[1] = (1 / v(2)) * v(1);
      out[2] = (1 / v(2));
      return;
    case LR::ANCHORED_INVERSE_DEPTH_SINGLE: {
      const double *b = getfej ? uv_norm_zero_fej : uv_norm_zero;
      for (int k = 0; k < 3; ++k) out[k] = (1.0 / v(0)) * b[k];
      return;
    }
    default:
      assert(false);
    }
  }
  void set_from_xyz(const double p[3], bool isfej) {
    typedef LandmarkRepresentation LR;
    VectorXd v(size(), 1);
    switch (_feat_representation) {
    case LR::GLOBAL_3D:
    case LR::ANCHORED_3D:
      for (int k = 0; k < 3; ++k) v(k) = p[k];
      break;
    case LR::GLOBAL_FULL_INVERSE_DEPTH:
    case LR::ANCHORED_FULL_INVERSE_DEPTH: {
      const double rho = 1 / std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
      v(0) = std::atan2(p[1], p[0]);
      v(1) = std::acos(rho * p[2]);
      v(2) = rho;
      break;
    }
    case LR::ANCHORED_MSCKF_INVERSE_DEPTH:
      v(0) = p[0] / p[2];
      v(1) = p[1] / p[2];
      v(2) = 1 / p[2];
      break;
    case LR::ANCHORED_INVERSE_DEPTH_SINGLE: {
      v(0) = 1.0 / p[2];
      double *b = isfej ? uv_norm_zero_fej : uv_norm_zero;
      for (int k = 0; k < 3; ++k) b[k] = (1.0 / p[2]) * p[k];
      break;
    }
    default:
      assert(false);
    }
    if (isfej) set_fej(v);
    else set_value(v);
  }
};

}  // namespace ov_type

namespace ov_core {
// subset of ext ov_core::Feature used by the updaters (mono: camera id 0)
struct Feature {
  size_t featid = 0;
  bool to_delete = false;
  std::vector<float> uvs;          // [2*k] raw pixels (reference: unordered_map<cam, vector<VectorXf>>; here flat, cam_ids says whose)
  std::vector<float> uvs_norm;     // [2*k] undistorted normalised coordinates (filled by the tracker); empty = p_FinG is given
  std::vector<double> timestamps;  // [k] clone timestamps
  // [k] camera of every measurement (reference: the key of the per-camera maps uvs / timestamps); empty = all of camera 0.  A
  // feature seen by two cameras at one clone time carries two measurements with the same timestamp.
  std::vector<int> cam_ids;
  int cam_of(size_t k) const { return k < cam_ids.size() ? cam_ids[k] : 0; }
  bool only_camera0() const {
    for (int c : cam_ids)
      if (c != 0) return false;
    return true;
  }
  double p_FinG[3] = {0, 0, 0};
  // anchor of the triangulated position (ext FeatureInitializer::single_triangulation: the last pose of the camera that saw
  // the feature most); -1 = triangulation has not filled it, p_FinA is then derived from p_FinG where it is needed
  int anchor_cam_id = -1;
  double anchor_clone_timestamp = -1;
  double p_FinA[3] = {0, 0, 0};
};
// ext ov_core::FeatureDatabase (feat/FeatureDatabase.h), the part the update classes use: id -> track, single camera
class FeatureDatabase {
public:
  std::shared_ptr<Feature> get_feature(size_t id, bool remove = false) {
    auto it = features_idlookup.find(id);
    if (it == features_idlookup.end()) return nullptr;
    auto f = it->second;
    if (remove) features_idlookup.erase(it);
    return f;
  }
  void update_feature(size_t id, double timestamp, size_t /*cam_id*/, float u, float v, float u_n, float v_n) {
    auto &f = features_idlookup[id];
    if (!f) {
      f = std::make_shared<Feature>();
      f->featid = id;
    }
    f->uvs.push_back(u);
    f->uvs.push_back(v);
    f->uvs_norm.push_back(u_n);
    f->uvs_norm.push_back(v_n);
    f->timestamps.push_back(timestamp);
  }
  // every track with a measurement at exactly this time
  std::vector<std::shared_ptr<Feature>> features_containing(double timestamp, bool remove = false, bool skip_deleted = false) {
    std::vector<std::shared_ptr<Feature>> out;
    for (auto it = features_idlookup.begin(); it != features_idlookup.end();) {
      auto &f = it->second;
      if (skip_deleted && f->to_delete) {
        ++it;
        continue;
      }
      bool has = false;
      for (double t : f->timestamps)
        if (t == timestamp) {
          has = true;
          break;
        }
      if (has) {
        out.push_back(f);
        if (remove) {
          it = features_idlookup.erase(it);
          continue;
        }
      }
      ++it;
    }
    return out;
  }
  // drops the measurements taken at exactly this time; tracks left empty are flagged
  void cleanup_measurements_exact(double timestamp) {
    for (auto it = features_idlookup.begin(); it != features_idlookup.end();) {
      auto &f = it->second;
      for (size_t k = 0; k < f->timestamps.size();) {
        if (f->timestamps[k] == timestamp) {
          f->timestamps.erase(f->timestamps.begin() + k);
          f->uvs.erase(f->uvs.begin() + 2 * k, f->uvs.begin() + 2 * k + 2);
          if (f->uvs_norm.size() >= 2 * (k + 1)) f->uvs_norm.erase(f->uvs_norm.begin() + 2 * k, f->uvs_norm.begin() + 2 * k + 2);
        } else {
          ++k;
        }
      }
      if (f->timestamps.empty()) {
        f->to_delete = true;
        it = features_idlookup.erase(it);
      } else {
        ++it;
      }
    }
  }
  std::unordered_map<size_t, std::shared_ptr<Feature>> &get_internal_data() { return features_idlookup; }
  size_t size() const { return features_idlookup.size(); }

protected:
  std::unordered_map<size_t, std::shared_ptr<Feature>> features_idlookup;
};
// ext ov_core::FeatureHelper::compute_disparity (feat/FeatureHelper.h): pixel displacement of the tracks seen at both times
struct FeatureHelper {
  static void compute_disparity(std::shared_ptr<FeatureDatabase> db, double time0, double time1, double &disp_mean, double &disp_var,
                                int &total_feats) {
    std::vector<double> disparities;
    for (auto &feat : db->features_containing(time0, false, true)) {
      int i0 = -1, i1 = -1;
      for (size_t k = 0; k < feat->timestamps.size(); ++k) {
        if (i0 < 0 && feat->timestamps[k] == time0) i0 = (int)k;
        if (i1 < 0 && feat->timestamps[k] == time1) i1 = (int)k;
      }
      if (i0 < 0 || i1 < 0) continue;
      const float du = feat->uvs[2 * i1] - feat->uvs[2 * i0], dv = feat->uvs[2 * i1 + 1] - feat->uvs[2 * i0 + 1];
      disparities.push_back((double)std::sqrt(du * du + dv * dv));  // Vector2f::norm()
    }
    if (disparities.size() < 2) {
      disp_mean = -1;
      disp_var = -1;
      total_feats = 0;
      return;
    }
    disp_mean = 0;
    for (double d : disparities) disp_mean += d;
    disp_mean /= (double)disparities.size();
    disp_var = 0;
    for (double d : disparities) disp_var += (d - disp_mean) * (d - disp_mean);
    disp_var = std::sqrt(disp_var / (double)(disparities.size() - 1));
    total_feats = (int)disparities.size();
  }
};
// ext ov_core::FeatureInitializerOptions (feat/FeatureInitializerOptions.h), defaults of open_vins
struct FeatureInitializerOptions {
  bool triangulate_1d = false;  // depth along the anchor bearing only (single_triangulation_1d)
  bool refine_features = true;
  int max_runs = 5;
  double init_lamda = 1e-3, max_lamda = 1e10, min_dx = 1e-6, min_dcost = 1e-6, lam_mult = 10;
  double min_dist = 0.10, max_dist = 60, max_baseline = 40, max_cond_number = 10000;
};
// ext ov_core::ImuData (utils/sensor_data.h)
struct ImuData {
  double timestamp = 0.0;
  double wm[3] = {0, 0, 0};  // angular velocity (rad/s)
  double am[3] = {0, 0, 0};  // linear acceleration (m/s^2)
};
}  // namespace ov_core
