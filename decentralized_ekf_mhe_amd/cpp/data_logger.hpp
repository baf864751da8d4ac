// data_logger.hpp — writes the reference's experiment-log format, so the authors' plotting scripts keep
// working on runs of this estimator.
//
// Format (src/decentral_legged_est/include/decentral_legged_est/data_logger.hpp:53-65, 66-200, 262-290;
// rows registered by robotSub in src/decentral_legged_est/src/EstSub.cpp:99-119):
//   <dir>/<name>_Name.csv   one line per registered variable, in registration order:  name,type,len,\n
//                           type in {double, int, VectorXd, VectorXf, VectorXi, Quaterniond}
//   <dir>/<name>_Data       raw little-endian values appended by every spin_logging(), variables in
//                           registration order: VectorXd / Quaterniond (w x y z) as float64, VectorXf as
//                           float32, int and VectorXi converted to float32
// The estimator node logs 27 doubles per tick: pose(3) GT_v(3) v_body(3) x_MHE(9) p_vo_accmulate_(3)
// filter_euler_(3) gt_euler_(3).
//
// Same method names as the reference class (init / add_data / spin_logging / done_logging).  Differences:
// the directory is given explicitly (the reference prepends $HOME), pointers are registered with an explicit
// length instead of Eigen types, and a scalar double is written as float64 (the reference has no overload
// for it).  Header-only, no dependencies.
#pragma once
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

class Data_Logger {
public:
    Data_Logger() = default;
    Data_Logger(const std::string& FileName, const std::string& FileLocation) { init(FileName, FileLocation); }
    Data_Logger(const Data_Logger&) = delete;
    Data_Logger& operator=(const Data_Logger&) = delete;
    ~Data_Logger() { done_logging(); }

    void init(const std::string& FileName, const std::string& FileLocation) {
        done_logging();
        std::string base = FileLocation;
        if (!base.empty() && base.back() != '/') base += '/';
        data_ = std::fopen((base + FileName + "_Data").c_str(), "wb");
        names_ = std::fopen((base + FileName + "_Name.csv").c_str(), "w");
        if (!data_ || !names_) { done_logging(); throw std::runtime_error("Data_Logger: cannot create " + base + FileName + "_Data / _Name.csv"); }
    }
    void add_data(const double* p, const std::string& name = "some_double") { reg(p, 1, name, "double", F64); }
    void add_data(const int* p, const std::string& name = "some_int") { reg(p, 1, name, "int", I32); }
    void add_data_vectorXd(const double* p, unsigned len, const std::string& name) { reg(p, len, name, "VectorXd", F64); }
    void add_data_vectorXf(const float* p, unsigned len, const std::string& name) { reg(p, len, name, "VectorXf", F32); }
    void add_data_vectorXi(const int* p, unsigned len, const std::string& name) { reg(p, len, name, "VectorXi", I32); }
    // quaternion given as w x y z
    void add_data_quaternion(const double* wxyz, const std::string& name) { reg(wxyz, 4, name, "Quaterniond", F64); }
    void add_data(const std::vector<double>& v, const std::string& name = "some_vec") { add_data_vectorXd(v.data(), (unsigned)v.size(), name); }

    void spin_logging() {
        if (!data_) throw std::runtime_error("Data_Logger: init() first");
        for (const Var& v : vars_) {
            if (v.kind == F64) std::fwrite(v.p, sizeof(double), v.len, data_);
            else if (v.kind == F32) std::fwrite(v.p, sizeof(float), v.len, data_);
            else
                for (unsigned i = 0; i < v.len; ++i) {
                    float f = static_cast<float>(static_cast<const int*>(v.p)[i]);
                    std::fwrite(&f, sizeof(float), 1, data_);
                }
        }
    }
    void done_logging() {
        if (data_) std::fclose(data_);
        if (names_) std::fclose(names_);
        data_ = names_ = nullptr;
    }

private:
    enum Kind { F64, F32, I32 };
    struct Var { const void* p; unsigned len; Kind kind; };
    void reg(const void* p, unsigned len, const std::string& name, const char* type, Kind kind) {
        if (!names_) throw std::runtime_error("Data_Logger: init() first");
        vars_.push_back({p, len, kind});
        std::fprintf(names_, "%s,%s,%u,\n", name.c_str(), type, len);
        std::fflush(names_);
    }
    std::vector<Var> vars_;
    std::FILE* data_ = nullptr;
    std::FILE* names_ = nullptr;
};
