// oracle/shim/ParameterObject.h -- TEST INFRASTRUCTURE ONLY.
//
// Stand-in for the un-vendored GenericParameters dependency of the reference
// (InteractiveComputerGraphics/GenericParameters, pinned a4e2744e... in
// /root/reference/CMakeLists.txt:69-76).  It is fetched by git at configure
// time upstream and is absent from /root/reference, so the oracle build
// (oracle/Makefile) puts this header on the include path instead.
//
// It implements only the registry surface the reference's Simulation/ layer
// calls (call sites: Simulation/TimeStepController.cpp:38-73,
// Simulation/SimulationModel.cpp:128-268, Simulation/Simulation.cpp:59-66,
// Simulation/CollisionDetection.cpp:39-47, Simulation/TimeStep.cpp:36).
// No hot-path arithmetic lives in GenericParameters; this file contains none.
// Assumption: EnumParameter::addEnumValue hands out ids
// 0,1,2,... in call order.
#ifndef PBDX_ORACLE_PARAMETEROBJECT_SHIM_H
#define PBDX_ORACLE_PARAMETEROBJECT_SHIM_H

#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace GenParam
{
	class ParameterBase
	{
	public:
		enum DataTypes { INT8 = 0, INT16, INT32, UINT8, UINT16, UINT32, FLOAT, DOUBLE, ENUM, BOOL, FUNCTION, VEC_FLOAT, VEC_DOUBLE, VEC_INT32, VEC_UINT32, STRING, LIST, STRUCT };
		ParameterBase(const std::string &name, const std::string &label) :
			m_name(name), m_label(label), m_readOnly(false), m_visible(true) {}
		virtual ~ParameterBase() {}
		const std::string &getName() const { return m_name; }
		const std::string &getLabel() const { return m_label; }
		void setGroup(const std::string &g) { m_group = g; }
		const std::string &getGroup() const { return m_group; }
		void setDescription(const std::string &d) { m_description = d; }
		const std::string &getDescription() const { return m_description; }
		void setReadOnly(const bool v) { m_readOnly = v; }
		bool getReadOnly() const { return m_readOnly; }
		void setVisible(const bool v) { m_visible = v; }
		void setHotKey(const std::string &) {}
	protected:
		std::string m_name, m_label, m_group, m_description;
		bool m_readOnly, m_visible;
	};

	template <typename T>
	class NumericParameter : public ParameterBase
	{
	public:
		typedef std::function<T()> GetFunc;
		typedef std::function<void(T)> SetFunc;
		NumericParameter(const std::string &n, const std::string &l, T *ptr) :
			ParameterBase(n, l), m_hasMin(false), m_hasMax(false)
		{
			m_get = [ptr]() { return *ptr; };
			m_set = [ptr](T v) { *ptr = v; };
		}
		NumericParameter(const std::string &n, const std::string &l, GetFunc g, SetFunc s) :
			ParameterBase(n, l), m_get(g), m_set(s), m_hasMin(false), m_hasMax(false) {}
		void setMinValue(const T v) { m_min = v; m_hasMin = true; }
		void setMaxValue(const T v) { m_max = v; m_hasMax = true; }
		T getValue() const { return m_get(); }
		void setValue(const T v)
		{
			T w = v;
			if (m_hasMin && w < m_min) w = m_min;
			if (m_hasMax && w > m_max) w = m_max;
			if (m_set) m_set(w);
		}
	protected:
		GetFunc m_get; SetFunc m_set;
		T m_min, m_max; bool m_hasMin, m_hasMax;
	};
	typedef NumericParameter<float> FloatParameter;
	typedef NumericParameter<double> DoubleParameter;
	typedef NumericParameter<unsigned int> UInt32Parameter;
	typedef NumericParameter<int> Int32Parameter;

	class BoolParameter : public NumericParameter<bool>
	{
	public:
		using NumericParameter<bool>::NumericParameter;
	};

	class EnumParameter : public NumericParameter<int>
	{
	public:
		struct EnumValue { int id; std::string name; };
		using NumericParameter<int>::NumericParameter;
		void addEnumValue(const std::string &name, int &id)
		{
			id = (int)m_values.size();
			m_values.push_back({ id, name });
		}
		const std::vector<EnumValue> &getEnumValues() const { return m_values; }
	protected:
		std::vector<EnumValue> m_values;
	};

	template <typename T>
	class VectorParameter : public ParameterBase
	{
	public:
		VectorParameter(const std::string &n, const std::string &l, unsigned int dim, T *ptr) :
			ParameterBase(n, l), m_dim(dim), m_ptr(ptr) {}
		T *getValue() const { return m_ptr; }
		void setValue(T *v) { for (unsigned int i = 0; i < m_dim; i++) m_ptr[i] = v[i]; }
		unsigned int getDim() const { return m_dim; }
	protected:
		unsigned int m_dim; T *m_ptr;
	};

	class ParameterObject
	{
	public:
		ParameterObject() {}
		virtual ~ParameterObject() {}
		virtual void initParameters() {}

		unsigned int numParameters() const { return (unsigned int)m_parameters.size(); }
		ParameterBase *getParameter(const unsigned int i) { return m_parameters[i].get(); }
		const ParameterBase *getParameter(const unsigned int i) const { return m_parameters[i].get(); }

		void setGroup(const unsigned int i, const std::string &g) { m_parameters[i]->setGroup(g); }
		void setDescription(const unsigned int i, const std::string &d) { m_parameters[i]->setDescription(d); }
		void setReadOnly(const unsigned int i, const bool v) { m_parameters[i]->setReadOnly(v); }
		void setVisible(const unsigned int i, const bool v) { m_parameters[i]->setVisible(v); }
		void setHotKey(const unsigned int, const std::string &) {}

		template <typename T>
		int createNumericParameter(const std::string &n, const std::string &l, T *ptr)
		{ return add(new NumericParameter<T>(n, l, ptr)); }
		template <typename T>
		int createNumericParameter(const std::string &n, const std::string &l, std::function<T()> g, std::function<void(T)> s)
		{ return add(new NumericParameter<T>(n, l, g, s)); }
		int createBoolParameter(const std::string &n, const std::string &l, bool *ptr)
		{ return add(new BoolParameter(n, l, ptr)); }
		int createBoolParameter(const std::string &n, const std::string &l, std::function<bool()> g, std::function<void(bool)> s)
		{ return add(new BoolParameter(n, l, g, s)); }
		int createEnumParameter(const std::string &n, const std::string &l, int *ptr)
		{ return add(new EnumParameter(n, l, ptr)); }
		int createEnumParameter(const std::string &n, const std::string &l, std::function<int()> g, std::function<void(int)> s)
		{ return add(new EnumParameter(n, l, g, s)); }
		template <typename T>
		int createVectorParameter(const std::string &n, const std::string &l, unsigned int dim, T *ptr)
		{ return add(new VectorParameter<T>(n, l, dim, ptr)); }

		template <typename T>
		T getValue(const unsigned int i) const
		{ return static_cast<const NumericParameter<T>*>(m_parameters[i].get())->getValue(); }
		template <typename T>
		void setValue(const unsigned int i, const T v)
		{ static_cast<NumericParameter<T>*>(m_parameters[i].get())->setValue(v); }
		template <typename T>
		T *getVecValue(const unsigned int i) const
		{ return static_cast<const VectorParameter<T>*>(m_parameters[i].get())->getValue(); }
		template <typename T>
		void setVecValue(const unsigned int i, T *v)
		{ static_cast<VectorParameter<T>*>(m_parameters[i].get())->setValue(v); }

	protected:
		int add(ParameterBase *p)
		{
			m_parameters.push_back(std::unique_ptr<ParameterBase>(p));
			return (int)m_parameters.size() - 1;
		}
		std::vector<std::unique_ptr<ParameterBase>> m_parameters;
	};
}

#endif
